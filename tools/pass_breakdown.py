#!/usr/bin/env python3
"""Per-kernel HIP-event times of one FB15k-237-shaped ranking pass (library timers) + the whole pass: tools/pass_breakdown.py [workload] [Q]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from coper_amd import data as cdata
from coper_amd.models import ConvE
name = sys.argv[1] if len(sys.argv) > 1 else "fb15k237_cpg"
md = cdata.model_descriptors(name)
Q = int(sys.argv[2]) if len(sys.argv) > 2 else cdata.CONFIGS[name]["queries"]
p = cdata.synthetic_params(md, 0)
m = ConvE(md, device="cuda:0", score_mode=os.environ.get("COPER_MODE", "bf16x3")).load_parameters(p).prepare()
q = cdata.synthetic_queries(md, Q, seed=0)
dq = {k: torch.as_tensor(v).cuda() for k, v in q.items()}
nnz = len(q["filt_idx"])
m.reserve(Q, nnz)
def step():
    return m.rank_pass(dq["e1"], dq["rel"], dq["e2"], dq["filt_indptr"], dq["filt_idx"], filt_nnz=nnz, want_equal=False)
for _ in range(200):
    step()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
for a, b in ev:
    a.record(); step(); b.record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in ev)
print("%s %s Q=%d: pass median %.4f ms min %.4f" % (os.path.basename(os.environ.get("COPER_HIP_LIB", "default")), name, Q, ts[len(ts) // 2], ts[0]))
m.profile(True)
for k in ("group", "dense", "conv", "tail", "score_count", "band_exact"):
    m.profile_read(k)
for _ in range(30):
    step()
torch.cuda.synchronize()
tot = 0.0
for k in ("group", "dense", "conv", "tail", "score_count", "band_exact"):
    ms, n = m.profile_read(k)
    if n:
        print("   %-12s %.4f ms (%d launches)" % (k, ms / n, n)); tot += ms / n
print("   sum %.4f ms" % tot)
