#!/usr/bin/env python3
"""The FB15k-237-shaped pass with a skewed relation distribution (real test sets are: a few relations carry most triples) next to
the SURVEY's uniform one: tools/bench_skew.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from coper_amd import data as cdata
from coper_amd.models import ConvE
md = cdata.model_descriptors("fb15k237_cpg")
m = ConvE(md, device="cuda:0", score_mode="bf16x3").load_parameters(cdata.synthetic_params(md, 0)).prepare()
Q = 20480
q = cdata.synthetic_queries(md, Q, seed=0)
rng = np.random.default_rng(3)
Rf = md["num_rel"] // 2
for name, rel in (("uniform", q["rel"]), ("zipf s=1.0", None), ("zipf s=1.5", None)):
    if rel is None:
        s = float(name.split("=")[1])
        w = 1.0 / np.arange(1, Rf + 1) ** s
        rel = rng.choice(Rf, size=Q, p=w / w.sum()).astype(np.int64)
    d = {k: torch.as_tensor(v).cuda() for k, v in dict(q, rel=rel).items()}
    cnt = np.bincount(rel, minlength=Rf)
    tiles = int(np.ceil(cnt[cnt > 32] / 128).sum() + (cnt[(cnt > 0) & (cnt <= 32)] > 0).sum())
    for _ in range(5): m.rank_pass(d["e1"], d["rel"], d["e2"], d["filt_indptr"], d["filt_idx"], want_equal=False)
    m.profile(True)
    for k in ("dense", "score_count", "tail", "group"): m.profile_read(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): m.rank_pass(d["e1"], d["rel"], d["e2"], d["filt_indptr"], d["filt_idx"], want_equal=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
    ks = {k: (lambda a: a[0] / max(a[1], 1))(m.profile_read(k)) for k in ("group", "dense", "tail", "score_count")}
    m.profile(False)
    print("%-11s relations present %3d, largest %5d queries, %3d tiles: pass %.1f us  (group %.1f, encoder %.1f, tail %.1f, count %.1f)" % (
        name, (cnt > 0).sum(), cnt.max(), tiles, dt * 1e6, ks["group"] * 1e3, ks["dense"] * 1e3, ks["tail"] * 1e3, ks["score_count"] * 1e3))
