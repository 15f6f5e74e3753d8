#!/usr/bin/env python3
"""A/B timing of the encode kernels for a given build of the library (COPER_HIP_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from coper_amd import data as cdata
from coper_amd.models import ConvE
name = sys.argv[1] if len(sys.argv) > 1 else "fb15k237_cpg"
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 20480
md = cdata.model_descriptors(name)
p = cdata.synthetic_params(md, 0)
m = ConvE(md, device="cuda:0", score_mode=os.environ.get("COPER_MODE", "bf16x3")).load_parameters(p).prepare()
q = cdata.synthetic_queries(md, Q, seed=0)
dq = {k: torch.as_tensor(v).cuda() for k, v in q.items()}
for _ in range(3):
    m.encode(dq["e1"], dq["rel"])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
def _mark(kind):   # tools/power_probe.sh: time window of the measured loop
    if os.environ.get("COPER_PP_MARK"):
        import time
        with open(os.environ["COPER_PP_MARK"], "a") as f:
            f.write("MARK %.2f %s encoder (fused conv + dense) %s Q=%d\n" % (time.time(), kind, name, Q))
if reps > 400:
    torch.cuda.synchronize(); _mark("start")
    for i in range(reps):
        m.encode(dq["e1"], dq["rel"])
        if i % 1000 == 999:
            torch.cuda.synchronize()
    torch.cuda.synchronize(); _mark("end")
    reps = 20
m.profile(True); m.profile_read("dense"); m.profile_read("conv")
for i in range(reps):
    m.encode(dq["e1"], dq["rel"])
torch.cuda.synchronize()
ms, n = m.profile_read("dense"); ms2, n2 = m.profile_read("conv")
print("%s: dense avg %.4f ms, conv avg %.4f ms" % (os.environ.get("COPER_HIP_LIB", "default"), ms / max(n, 1), ms2 / max(n2, 1)))

try:      # -DCOPER_DBG_FUSED_CLOCK build: when each workgroup of the last launch started and ended, by tile size
    import ctypes
    import numpy as np
    lib = ctypes.CDLL(os.environ.get("COPER_HIP_LIB", ""))
    buf = (ctypes.c_ulonglong * (3 * 2048))()
    if lib.coper_dbg_fused_clock(buf, 2048) == 0:
        a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 3).astype(np.int64)
        a = a[a[:, 2] > 0]
        t0 = a[:, 1].min()
        st, en, n = (a[:, 1] - t0) * 0.01, (a[:, 2] - t0) * 0.01, a[:, 0]
        print("   %d workgroups; start: median %.1f us, last %.1f us; end: median %.1f, p90 %.1f, last %.1f us" % (
            len(a), np.median(st), st.max(), np.median(en), np.percentile(en, 90), en.max()))
        for lo, hi in ((1, 32), (33, 64), (65, 80), (81, 96), (97, 112), (113, 128)):
            m_ = (n >= lo) & (n <= hi)
            if m_.any():
                print("   tiles of %3d..%3d queries: %3d workgroups, duration median %.1f us, max %.1f us, end max %.1f us" % (
                    lo, hi, m_.sum(), np.median((en - st)[m_]), (en - st)[m_].max(), en[m_].max()))
except (OSError, AttributeError):
    pass
