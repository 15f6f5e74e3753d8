#!/usr/bin/env python3
"""A/B timing of the score_count kernel for a given build of the library (COPER_HIP_LIB=...)."""
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
h = m.encode(q["e1"], q["rel"])
tgt = m.target_scores(h, q["e2"])
dq = {k: torch.as_tensor(v).cuda() for k, v in q.items()}
for _ in range(3):
    m.rank_counts(h, tgt, dq["e2"], dq["filt_indptr"], dq["filt_idx"])
m.profile(True); m.profile_read("score_count")
for _ in range(40):
    m.rank_counts(h, tgt, dq["e2"], dq["filt_indptr"], dq["filt_idx"])
torch.cuda.synchronize()
ms, n = m.profile_read("score_count")
fl = 2.0 * Q * md["num_ent"] * md["ent_emb_size"]
print("%s: score_count avg %.4f ms  %.1f TFLOP/s (%.1f%% of 157.3)" % (os.environ.get("COPER_HIP_LIB", "default"), ms / n, fl / (ms / n * 1e-3) / 1e12, fl / (ms / n * 1e-3) / 1e12 / 1.573))
