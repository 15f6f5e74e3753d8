#!/usr/bin/env python3
"""Latency of ONE reference-sized evaluation batch (512 queries) replayed from a hipGraph, and of eager passes of a few sizes:
tools/bench_batch.py [workload]   (COPER_HIP_LIB selects an A/B build)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from coper_amd import data as cdata
from coper_amd.models import ConvE
name = sys.argv[1] if len(sys.argv) > 1 else "fb15k237_cpg"
md = cdata.model_descriptors(name)
m = ConvE(md, device="cuda:0", score_mode="bf16x3").load_parameters(cdata.synthetic_params(md, 0)).prepare()
q = cdata.synthetic_queries(md, 512, seed=0)
run = m.capture_rank_pass(512, int(len(q["filt_idx"]) * 2))
dq = {k: torch.as_tensor(v).cuda() for k, v in q.items()}
for _ in range(20): run(dq["e1"], dq["rel"], dq["e2"], dq["filt_indptr"], dq["filt_idx"])
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): run(dq["e1"], dq["rel"], dq["e2"], dq["filt_indptr"], dq["filt_idx"])
torch.cuda.synchronize()
print("%s: hipGraph batch of 512: %.1f us" % (os.path.basename(os.environ.get("COPER_HIP_LIB", "default")), (time.perf_counter() - t0) / 200 * 1e6))
for Q in (512, 2048, 4096, 20480):
    qq = cdata.synthetic_queries(md, Q, seed=1)
    d = {k: torch.as_tensor(v).cuda() for k, v in qq.items()}
    for _ in range(5): m.rank_pass(d["e1"], d["rel"], d["e2"], d["filt_indptr"], d["filt_idx"], want_equal=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): m.rank_pass(d["e1"], d["rel"], d["e2"], d["filt_indptr"], d["filt_idx"], want_equal=False)
    torch.cuda.synchronize()
    print("   eager pass of %5d queries: %.1f us" % (Q, (time.perf_counter() - t0) / 50 * 1e6))
