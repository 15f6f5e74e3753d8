#!/usr/bin/env python3
"""Timing of the fused tail (finalize + targets + filter correction) inside ranking passes; with a COPER_DBG_TL_CLOCK build
(tools/ab_build.py tlclk=-DCOPER_DBG_TL_CLOCK) also the phase boundaries inside the kernel.  tools/ab_tail.py [workload] [Q]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from coper_amd import data as cdata
from coper_amd.models import ConvE
name = sys.argv[1] if len(sys.argv) > 1 else "fb15k237_cpg"
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 20480
md = cdata.model_descriptors(name)
m = ConvE(md, device="cuda:0", score_mode="bf16x3").load_parameters(cdata.synthetic_params(md, 0)).prepare()
q = cdata.synthetic_queries(md, Q, seed=0)
dq = {k: torch.as_tensor(v).cuda() for k, v in q.items()}
for _ in range(5):
    m.rank_pass(dq["e1"], dq["rel"], dq["e2"], dq["filt_indptr"], dq["filt_idx"], want_equal=False)
m.profile(True)
for k in ("tail", "score_count", "dense", "group"): m.profile_read(k)
for _ in range(40):
    m.rank_pass(dq["e1"], dq["rel"], dq["e2"], dq["filt_indptr"], dq["filt_idx"], want_equal=False)
torch.cuda.synchronize()
for k in ("group", "dense", "tail", "score_count"):
    ms, n = m.profile_read(k)
    if n: print("%-12s avg %.4f ms (%d)" % (k, ms / n, n))
try:
    lib = ctypes.CDLL(os.environ.get("COPER_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "coper_amd", "libcoper_hip.so")))
    out = (ctypes.c_double * 12)()
    if lib.coper_dbg_tl_clock((Q + 31) // 32, out) == 0:
        names = ["start", "finalize done", "fragments exchanged", "targets done", "filter tiles done", "end"]
        for i, nm in enumerate(names):
            print("   %-22s median %6.2f us   last workgroup %6.2f us" % (nm, out[i], out[6 + i]))
except AttributeError:
    pass
