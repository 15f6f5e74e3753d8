#!/usr/bin/env python3
"""A/B timing of the score_count kernel for a given build of the library (COPER_HIP_LIB=...).
tools/ab_score.py [workload] [Q] -- the 10M-entity config draws its table on the device."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from coper_amd import data as cdata
from coper_amd.models import ConvE
name = sys.argv[1] if len(sys.argv) > 1 else "fb15k237_cpg"
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 20480
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
md = cdata.model_descriptors(name)
big = md["num_ent"] * md["ent_emb_size"] > (1 << 28)
p = cdata.synthetic_params(md, 0, skip=("ent_emb", "pred_bias") if big else ())
if big:
    p["ent_emb"], p["pred_bias"] = cdata.synthetic_entity_rows_device(md, 0, "cuda:0")
m = ConvE(md, device="cuda:0", score_mode=os.environ.get("COPER_MODE", "bf16x3")).load_parameters(p).prepare()
q = cdata.synthetic_queries(md, Q, seed=0)
h = m.encode(q["e1"], q["rel"])
tgt = m.target_scores(h, q["e2"])
dq = {k: torch.as_tensor(v).cuda() for k, v in q.items()}
for _ in range(3):
    m.rank_counts(h, tgt, dq["e2"], dq["filt_indptr"], dq["filt_idx"])
def _mark(kind):   # tools/power_probe.sh: time window of the measured loop
    if os.environ.get("COPER_PP_MARK"):
        import time
        with open(os.environ["COPER_PP_MARK"], "a") as f:
            f.write("MARK %.2f %s count kernel %s Q=%d\n" % (time.time(), kind, name, Q))
if reps > 400:      # a long unprofiled loop for the power sampler, then the usual 40 timed launches
    torch.cuda.synchronize(); _mark("start")
    for i in range(reps):
        m.rank_counts(h, tgt, dq["e2"], dq["filt_indptr"], dq["filt_idx"])
        if i % 1000 == 999:
            torch.cuda.synchronize()       # bound the launch queue
    torch.cuda.synchronize(); _mark("end")
    reps = 40
m.profile(True); m.profile_read("score_count")
for i in range(reps):
    ng, ne = m.rank_counts(h, tgt, dq["e2"], dq["filt_indptr"], dq["filt_idx"])
torch.cuda.synchronize()
ms, n = m.profile_read("score_count")
t = ms / n
fl = 2.0 * Q * md["num_ent"] * md["ent_emb_size"]
by = md["num_ent"] * md["ent_emb_size"] * 4.0
print("%s %s Q=%d: score_count avg %.4f ms  %.1f TFLOP/s  %.2f TB/s (table once)  sum(ng)=%d" % (
    os.path.basename(os.environ.get("COPER_HIP_LIB", "default")), name, Q, t, fl / (t * 1e-3) / 1e12, by / (t * 1e-3) / 1e12,
    int(ng.sum().item())))

try:
    import ctypes
    lib = ctypes.CDLL(os.environ.get("COPER_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "coper_amd", "libcoper_hip.so")))
    g, u = ctypes.c_double(), ctypes.c_double()
    if lib.coper_dbg_clock(256, ctypes.byref(g), ctypes.byref(u)) == 0:
        print("   in-kernel clock (median over workgroups): %.3f GHz, in-kernel time %.1f us" % (g.value, u.value))
except AttributeError:
    pass
