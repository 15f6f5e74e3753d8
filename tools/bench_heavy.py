"""Cost of filter rows with thousands of known answers in coper_encode_rank (ranks only, x3 mode), FB15k-237 shapes:
the pass with the synthetic filter as it is, with one / ten / a hundred queries holding 5,000 more known answers each, and
the two-call path (coper_encode + coper_rank) on the same batches.  python tools/bench_heavy.py [--queries 20480]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coper_amd import data as cdata   # noqa: E402
from coper_amd.models import ConvE    # noqa: E402


def timed(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2], t[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=20480)
    ap.add_argument("--workload", default="fb15k237_cpg")
    a = ap.parse_args()
    md = cdata.model_descriptors(a.workload)
    m = ConvE(md, device="cuda:0", score_mode="bf16x3").load_parameters(cdata.synthetic_params(md, 0)).prepare()
    E, Q = md["num_ent"], a.queries
    q = cdata.synthetic_queries(md, Q, seed=0)
    rng = np.random.default_rng(1)
    base_rows = [q["filt_idx"][q["filt_indptr"][i]:q["filt_indptr"][i + 1]] for i in range(Q)]
    dev = lambda x: torch.as_tensor(x).to("cuda:0")
    e1, rel, e2 = dev(q["e1"]), dev(q["rel"]), dev(q["e2"])
    for n_heavy in (0, 1, 10, 100):
        rows = list(base_rows)
        for i in rng.choice(Q, n_heavy, replace=False):
            rows[i] = np.unique(np.concatenate([rows[i], rng.choice(E, 5000, replace=False)]))
        indptr = np.zeros(Q + 1, np.int64)
        indptr[1:] = np.cumsum([len(r) for r in rows])
        idx = np.concatenate(rows)
        ip, ix, nnz = dev(indptr), dev(idx), int(indptr[-1])
        fused = timed(lambda: m.rank_pass(e1, rel, e2, ip, ix, filt_nnz=nnz, want_equal=False))
        two = timed(lambda: m.rank(m.encode(e1, rel), e2, ip, ix, filt_nnz=nnz))
        r1 = m.rank_pass(e1, rel, e2, ip, ix, filt_nnz=nnz, want_equal=False)[0]
        r2 = m.rank(m.encode(e1, rel), e2, ip, ix, filt_nnz=nnz)[0]
        print("queries with +5000 known answers: %3d  nnz %8d | coper_encode_rank median %.4f ms (min %.4f) | encode + rank %.4f ms (min %.4f) | same ranks %s"
              % (n_heavy, nnz, fused[0], fused[1], two[0], two[1], bool(torch.equal(r1, r2))), flush=True)


if __name__ == "__main__":
    main()
