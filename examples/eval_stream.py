#!/usr/bin/env python3
"""A stream of evaluation passes with the transfers off the critical path: the pattern `bench.py --h2d overlap` times.

Every pass scores one batch of (e1, rel, e2, known answers) queries against all entities (what `ranking_and_hits` does per
`session.run`, `CoPER_ConvE/qa_cpg/metrics.py:38-57`).  The host keeps each batch as int32 in ONE pinned buffer; while pass n
runs, extra workgroups of its encoder launch read pass n + 1's batch over PCIe (`ConvE.stage_next` = `coper_stage_ids_next`),
one more sorts it by relation (`ConvE.group_next` = `coper_group_next`: pass n + 1 starts with its encoder launch), and pass n's
ranks leave for pinned host memory beside pass n + 1's first launch (`ConvE.post_next` = `coper_post_i32_next`).
One stream, no copy engine, no launch that only moves data -- except in front of the first pass and behind the last one.
The pipeline itself is `coper_amd.stream.RankStream` (round 6), which also acts on the guard of a sorting made ahead: a pass whose
staged ids were rewritten behind its sorting comes back as COPER_RANK_STALE and is ranked again.

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
from coper_amd.stream import RankStream  # noqa: E402

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
    # the pipeline (coper_amd/stream.py): two staging arrays, two rank buffers, three registrations per pass
    stream = RankStream(model, Q, max(len(b["filt_idx"]) for b in batches))
    packed = [stream.pack(b) for b in batches]                      # a host that scores the same batches again packs them once
    stream.run(packed)                                              # (warm-up: code objects, workspaces, the second set of grouping arrays)
    t0 = time.perf_counter()
    ranks = stream.run(packed)
    dt = time.perf_counter() - t0
    assert stream.stale_passes == 0                                 # (nothing rewrote a staged batch behind its sorting: see the guard)

    for n, b in enumerate(batches):                                 # the same ranks as one pass at a time, nothing overlapped
        r, _ = model.rank_pass(b["e1"], b["rel"], b["e2"], b["filt_indptr"], b["filt_idx"], want_equal=False)
        assert np.array_equal(ranks[n], r.cpu().numpy()), n
    rk = np.concatenate(ranks).astype(np.int64)
    print("%d passes of %d queries in %.3f ms (%.2f M scored triples/s, the first batch's own transfer and the last ranks' included); MRR %.4f, Hits@10 %.4f" % (
        args.batches, Q, dt * 1e3, args.batches * Q / dt / 1e6, float(np.mean(1.0 / rk)), float(np.mean(rk <= 10))))
    model.close()


if __name__ == "__main__":
    main()
