#!/usr/bin/env python3
"""Where a workgroup of the fused encoder spends its time (a build with -DCOPER_DBG_FUSED_CLOCK):
   python tools/ab_build.py --only kernels_dense_fused_bf16.hip fclk=-DCOPER_DBG_FUSED_CLOCK
   COPER_HIP_LIB=build/ab/lib_fclk.so python tools/fused_phases.py [workload]"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
from coper_amd import data as cdata
from coper_amd.models import ConvE
name = sys.argv[1] if len(sys.argv) > 1 else "wn18rr_cpg"
md = cdata.model_descriptors(name)
Q = cdata.CONFIGS[name]["queries"]
m = ConvE(md, device="cuda:0", score_mode="bf16x3").load_parameters(cdata.synthetic_params(md, 0)).prepare()
q = cdata.synthetic_queries(md, Q, seed=1)
dq = {k: torch.as_tensor(np.asarray(v)).to("cuda:0") for k, v in q.items()}
for i in range(6):
    m.rank_pass(dq["e1"], dq["rel"], dq["e2"], dq["filt_indptr"], dq["filt_idx"], filt_nnz=len(q["filt_idx"]), want_equal=False)
torch.cuda.synchronize()
N = 2048
out = (ctypes.c_ulonglong * (3 * N))()
assert m._lib.coper_dbg_fused_clock(out, N) == 0
a = np.array(out, dtype=np.uint64).reshape(N, 3).astype(np.int64)
live = a[a[:, 1] > 0]
live = live[live[:, 0] > 0] if (live[:, 0] > 0).any() else live
t0 = live[:, 1].min()
st = (live[:, 1] - t0) / 100.0; en = (live[:, 2] - t0) / 100.0
print(name, "workgroups with a tile:", len(live), "launch span %.1f us" % en.max())
print("start: median %.1f max %.1f | duration: min %.1f median %.1f max %.1f | end: median %.1f" % (np.median(st), st.max(), (en - st).min(), np.median(en - st), (en - st).max(), np.median(en)))
order = np.argsort(st)
for j in order[:: max(1, len(order) // 12)]:
    print("  n=%3d start %6.1f dur %6.1f" % (live[j, 0], st[j], en[j] - st[j]))
ph = (ctypes.c_ulonglong * (8 * N))()
if hasattr(m._lib, "coper_dbg_fused_phases") and m._lib.coper_dbg_fused_phases(ph, N) == 0:
    p = np.array(ph, dtype=np.uint64).reshape(N, 8).astype(np.int64)
    sel = (a[:, 1] > 0) & (a[:, 0] > 0) & (p[:, 3] > 0)
    A0 = a[sel]; P0 = p[sel]
    rel = lambda col: (P0[:, col] - A0[:, 1]) / 100.0
    print("phases (us after the workgroup's start, medians): weights issued %.1f | images in LDS %.1f (conv role sees them %.1f) | first 2P steps done %.1f | loop done %.1f | end %.1f"
          % (np.median(rel(0)), np.median(rel(1)), np.median(rel(5)), np.median(rel(2)), np.median(rel(3)), np.median((A0[:, 2] - A0[:, 1]) / 100.0)))
