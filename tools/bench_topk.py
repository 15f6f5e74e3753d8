#!/usr/bin/env python3
"""Timing of coper_rank_counts with and without top-k (k = 0, 10, 32), bf16x3: FB15k-237-shaped, WN18RR-shaped, and one
1.25 M-entity shard of the 10M-entity config (random h; the kernels do not care)."""
import sys, time
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
import torch
from coper_amd import data as cdata
from coper_amd.models import ConvE
def run(name, Q, shard=None, big=False):
    md = cdata.model_descriptors(name)
    if big:
        import bench
        dev = torch.device("cuda", 0)
        params, _ = bench.device_params(md, 0, dev, shard)
        m = ConvE(md, device=dev, shard=shard, score_mode="bf16x3")
        m.load_parameters(params, global_rows=False); m.prepare()
    else:
        p = cdata.synthetic_params(md, 0)
        m = ConvE(md, device="cuda:0", score_mode="bf16x3").load_parameters(p).prepare()
    q = cdata.synthetic_queries(md, Q, seed=0)
    if shard:
        h = torch.randn(Q, md["ent_emb_size"], device="cuda").abs() * 0.3
        tgt = torch.zeros((2, Q), device="cuda")
    else:
        h = m.encode(q["e1"], q["rel"])
        tgt = m.target_scores(h, q["e2"])
    dq = {k: torch.as_tensor(v).cuda() for k, v in q.items()}
    for k in (0, 10, 32):
        for _ in range(2): m.rank_counts(h, tgt, dq["e2"], dq["filt_indptr"], dq["filt_idx"], k=k)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): m.rank_counts(h, tgt, dq["e2"], dq["filt_indptr"], dq["filt_idx"], k=k)
        torch.cuda.synchronize()
        print(name, shard, "k=%d" % k, "%.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
        try:      # -DCOPER_DBG_TK_OVER build: strips of the threshold kernel that left the fast path (7 launches above)
            import ctypes, os
            lib = ctypes.CDLL(os.environ.get("COPER_HIP_LIB", ""))
            print("    strips on the general route: %d of %d" % (lib.coper_dbg_tk_over(), 7 * ((Q + 15) // 16)))
        except (OSError, AttributeError):
            pass
    m.close()
run("fb15k237_cpg", 20480)
run("wn18rr_cpg", 3072)
run("synth10m_cpg", 4096, shard=(0, 1250000), big=True)
