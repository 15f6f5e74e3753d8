#!/usr/bin/env python3
"""A stream of evaluation passes with the transfers off the critical path: the pattern `bench.py --h2d overlap` times.

Every pass scores one batch of (e1, rel, e2, known answers) queries against all entities (what `ranking_and_hits` does per
`session.run`, `CoPER_ConvE/qa_cpg/metrics.py:38-57`).  The host keeps each batch as int32 in ONE pinned buffer; while pass n
runs, extra workgroups of its encoder launch read pass n + 1's batch over PCIe (`ConvE.stage_next` = `coper_stage_ids_next`),
one more sorts it by relation (`ConvE.group_next` = `coper_group_next`: pass n + 1 starts with its encoder launch), and pass n's
ranks leave for pinned host memory beside pass n + 1's first launch (`ConvE.post_next` = `coper_post_i32_next`).
One stream, no copy engine, no launch that only moves data -- except in front of the first pass and behind the last one.

    python examples/eval_stream.py [--workload fb15k237_cpg] [--batches 6] [--queries 20480] [--score-mode bf16x3|f32]"""
import argparse
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from coper_amd import data as cdata  # noqa: E402
from coper_amd.models import ConvE  # noqa: E402

KEYS = ("e1", "rel", "e2", "filt_indptr", "filt_idx")


def pack(q, nnz_cap):
    """[e1 | rel | e2 | filt_indptr | filt_idx] as int32 in one pinned buffer (filt_idx padded to a common capacity, so that every
    batch of the stream has the same layout), and the offsets of the five arrays."""
    sizes = [len(q["e1"]), len(q["rel"]), len(q["e2"]), len(q["filt_indptr"]), nnz_cap]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    pin = torch.zeros(int(offs[-1]), dtype=torch.int32).pin_memory()
    for k, o in zip(KEYS, offs):
        a = np.asarray(q[k])
        pin[o:o + len(a)] = torch.as_tensor(a.astype(np.int32))
    return pin, offs, sizes


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="fb15k237_cpg")
    ap.add_argument("--batches", type=int, default=6)
    ap.add_argument("--queries", type=int, default=None)
    ap.add_argument("--score-mode", choices=["f32", "bf16x3"], default="bf16x3", help="bf16x3: the headline mode (ranks of the fp32 chain); f32: bit-exact logits")
    args = ap.parse_args(argv)
    md = cdata.model_descriptors(args.workload)
    Q = args.queries or cdata.CONFIGS[args.workload]["queries"]
    dev = torch.device("cuda", 0)
    model = ConvE(md, device=dev, score_mode=args.score_mode).load_parameters(cdata.synthetic_params(md, 0)).prepare()
    batches = [cdata.synthetic_queries(md, Q, seed=s) for s in range(args.batches)]
    cap = max(len(b["filt_idx"]) for b in batches)
    packed = [pack(b, cap) for b in batches]
    offs, sizes = packed[0][1], packed[0][2]
    # two staging arrays (the pass that runs reads one while the next batch arrives in the other), two rank buffers
    stages = [torch.empty(int(offs[-1]), dtype=torch.int64, device=dev) for _ in range(2)]
    views = [{k: st[o:o + n] for k, o, n in zip(KEYS, offs, sizes)} for st in stages]
    ranks_dev = [torch.empty(Q, dtype=torch.int32, device=dev) for _ in range(2)]
    ranks_host = [torch.empty(Q, dtype=torch.int32).pin_memory() for _ in range(args.batches)]
    model.reserve(Q, cap)
    b0 = batches[0]
    model.rank_pass(b0["e1"], b0["rel"], b0["e2"], b0["filt_indptr"], b0["filt_idx"], want_equal=False)    # (warm-up: code objects, workspaces)

    def stream():
        model.widen_ids(packed[0][0], out=stages[0])                   # the first batch has no pass to arrive beside
        for n in range(args.batches):
            c = n & 1
            if n + 1 < args.batches:
                model.stage_next(packed[n + 1][0], stages[1 - c])          # batch n + 1: read beside this pass's encoder launch
                model.group_next(views[1 - c]["e1"], views[1 - c]["rel"])  # ... and sorted by relation there: pass n + 1 starts with its encoder
            v = views[c]
            nnz = len(batches[n]["filt_idx"])
            r, _ = model.rank_pass(v["e1"], v["rel"], v["e2"], v["filt_indptr"], v["filt_idx"][:nnz], filt_nnz=nnz, want_equal=False,
                                   out=ranks_dev[c])
            model.post_next(r, ranks_host[n])                          # ranks n: out beside pass n + 1's first launch
        model.post_flush()                                             # ... and the last ones by a launch of their own
        torch.cuda.synchronize(dev)

    stream()                                                           # (the first run allocates the second set of grouping arrays)
    for h in ranks_host:
        h.fill_(-1)
    t0 = time.perf_counter()
    stream()
    dt = time.perf_counter() - t0

    for n, b in enumerate(batches):                                    # the same ranks as one pass at a time, nothing overlapped
        r, _ = model.rank_pass(b["e1"], b["rel"], b["e2"], b["filt_indptr"], b["filt_idx"], want_equal=False)
        assert np.array_equal(ranks_host[n].numpy(), r.cpu().numpy()), n
    rk = np.concatenate([h.numpy() for h in ranks_host]).astype(np.int64)
    print("%d passes of %d queries in %.3f ms (%.2f M scored triples/s, the first batch's own transfer and the last ranks' included); MRR %.4f, Hits@10 %.4f" % (
        args.batches, Q, dt * 1e3, args.batches * Q / dt / 1e6, float(np.mean(1.0 / rk)), float(np.mean(rk <= 10))))
    model.close()


if __name__ == "__main__":
    main()
