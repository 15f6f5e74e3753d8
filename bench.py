#!/usr/bin/env python3
"""bench.py -- scored triples/sec (1-vs-all) of the CoPER-ConvE evaluation hot path on MI355X.

A "step" = one pass of the hot path (encode -> 1-vs-all score -> filtered rank) over one batch of
synthetic queries: the evaluation set of the workload (Q queries), resident in
HBM before the timed region starts.  Default workload = BASELINE.json configs[1]
(FB15k-237-shaped CoPER-ConvE: |E|=14541, R2=474, d=200, r=32; Q=20480), fp32-exact mode.

  python bench.py --gpus N --steps K --warmup W
  N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
         query-sharded (every rank holds the model and ranks its own Q queries): weak scaling.
  --workload synth10m_cpg --mode entity : the 10M-entity config, entity-sharded (strong scaling).

Prints ONE JSON line (rank 0)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix)
PEAK_HBM_GBS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA; the bf16x3 mode spends 3 hardware MFMAs per algorithmic product


def pmc_traffic(workload, Q, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary (profiles/), collected on
    this same command in separate --pmc passes; None when no summary matches the workload."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload") != workload or d.get("queries") != Q:
            continue
        for name, v in d.get("kernels", {}).items():     # template arguments (<13>, <false>) follow the base name
            if name == kernel or name.startswith(kernel + "<"):
                return v["hbm_bytes_per_launch"]
    return None


def pmc_mfma_busy(workload, Q, kernel):
    """Matrix-pipe busy fraction and effective clock of `kernel` from the committed PMC summary
    (profiles/*pmc_mfma_busy*.json: SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE on this same command), or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_mfma_busy*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload") != workload or d.get("queries") != Q:
            continue
        for name, v in d.get("kernels", {}).items():
            if name == kernel or name.startswith(kernel + "<"):
                return {"mfma_busy_frac": v["mfma_busy_frac"], "effective_clock_ghz": v["effective_clock_ghz"]}
    return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="fb15k237_cpg")
    ap.add_argument("--mode", choices=["query", "entity"], default="query")
    ap.add_argument("--queries", type=int, default=None)
    ap.add_argument("--order", choices=["shuffled", "sorted"], default="shuffled")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--score-mode", choices=["f32", "bf16x3"], default="bf16x3")
    ap.add_argument("--profile-every", type=int, default=4,
                    help="HIP-event timing of the dominant kernels on every N-th timed step (an event pair around a launch "
                         "costs the stream a few microseconds of pipeline drain; 1 = every step)")
    ap.add_argument("--topk", type=int, default=0,
                    help="entity mode: also select and exchange the per-shard top-k of the filtered rows (SURVEY 8(e) step 3)")
    ap.add_argument("--dist-backend", default="nccl",
                    help="torch.distributed backend (nccl = RCCL).  gloo + several ranks on one GPU is a debugging aid "
                         "for the multi-process path on a single-GPU box; its numbers mean nothing")
    return ap.parse_args()


def device_params(md, seed, device, shard=None):
    """Random-init weights of the named architecture.  Small tensors come from the seeded numpy
    generator; an entity table too large to build on the host is drawn on the device (same
    N(0, 0.3^2) law), row-sharded."""
    import torch
    from coper_amd import data as cdata
    big = int(md["num_ent"]) * int(md["ent_emb_size"]) > (1 << 28)
    p = cdata.synthetic_params(md, seed, skip=("ent_emb", "pred_bias") if big else ())
    if big:
        lo, hi = shard if shard is not None else (0, int(md["num_ent"]))
        g = torch.Generator(device=device)
        g.manual_seed(seed * 1000 + lo)
        p["ent_emb"] = torch.randn((hi - lo, int(md["ent_emb_size"])), generator=g, device=device, dtype=torch.float32) * 0.3
        p["pred_bias"] = torch.randn((hi - lo,), generator=g, device=device, dtype=torch.float32) * 0.1
    return p, big


def cpu_baseline(md, params, q, seconds):
    """Reference-semantics CPU restatement (oracle, kind "port"): forward with the generated dense
    weights materialised [B,F,d] (models.py:70,412), logits for all entities, dense mask, per-row
    np.argsort (metrics.py:44-57), on a bounded sample of the same workload."""
    from oracle import coper_oracle as O
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    bs = 256
    done, t0 = 0, time.perf_counter()
    Q = len(q["e1"])
    while done + bs <= Q:
        ip = q["filt_indptr"][done:done + bs + 1]
        O.eval_pass_reference_semantics(params, md, q["e1"][done:done + bs], q["rel"][done:done + bs],
                                        q["e2"][done:done + bs], ip - ip[0], q["filt_idx"][ip[0]:ip[-1]], batch_size=bs)
        done += bs
        if time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "triples/s", "cores": int(threads), "kind": "port",
            "sample": "%d queries of the same workload in batches of %d (%.1f s): NumPy forward with materialised "
                      "[B,F,d] generated weights + dense mask + per-row argsort" % (done, bs, dt)}


def main():
    args = parse_args()
    # stdout carries exactly one line, the JSON result of rank 0: whatever libraries print on the way (RCCL's version
    # banner at communicator creation goes to stdout) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from coper_amd import data as cdata
    from coper_amd.models import ConvE
    from coper_amd.sharding import EntityShardedRanker, local_rank_pass, shard_bounds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    if args.dist_backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())   # debugging: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or bool(os.environ.get("COPER_BENCH_FORCE_DIST"))   # the env var: exercise the RCCL path with one rank
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.dist_backend)

    md = cdata.model_descriptors(args.workload)
    Q = args.queries or cdata.CONFIGS[args.workload]["queries"]
    entity_mode = args.mode == "entity"
    shard = shard_bounds(md["num_ent"], world, rank) if entity_mode else None
    params, big = device_params(md, 0, device, shard)
    model = ConvE(md, device=device, shard=shard, score_mode=args.score_mode)
    model.load_parameters(params, global_rows=not big)
    t_prep0 = time.perf_counter()
    model.prepare()
    torch.cuda.synchronize(device)
    prepare_ms = (time.perf_counter() - t_prep0) * 1e3

    # queries: entity mode -> every rank sees the same Q queries; query mode -> Q per rank (weak scaling)
    q = cdata.synthetic_queries(md, Q, seed=0 if entity_mode else rank, order=args.order)
    dev_q = {k: torch.as_tensor(v).to(device) for k, v in q.items()}   # inputs resident in HBM
    nnz = int(len(q["filt_idx"]))
    model.reserve(Q, nnz)
    ranker = EntityShardedRanker(model) if entity_mode else None
    host_q = dict(q)

    def step():
        if entity_mode:
            # ids are tiny; the host copies drive the relation split, the device copies feed the kernels
            return ranker.rank(dict(e1=host_q["e1"], rel=host_q["rel"], e2=dev_q["e2"], filt_indptr=dev_q["filt_indptr"],
                                    filt_idx=dev_q["filt_idx"]), k=args.topk)[:2]
        # one call per pass, like one session.run of the reference's ranker loop; like ranking_and_hits (and the
        # reference) the pass needs ranks only: tie counts and the embedding itself are not requested
        return model.rank_pass(dev_q["e1"], dev_q["rel"], dev_q["e2"], dev_q["filt_indptr"], dev_q["filt_idx"], filt_nnz=nnz,
                               want_equal=False)

    for _ in range(args.warmup):
        step()
    model.profile(True)
    for k in ("score_count", "dense", "conv"):
        model.profile_read(k)
    model.profile(False)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(args.steps):
        model.profile(i % max(1, args.profile_every) == 0)     # per-kernel HIP events on a sample of the timed steps
        ranks, _ = step()
    torch.cuda.synchronize(device)
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    model.profile(False)
    if use_dist:
        tmax = torch.tensor([dt], device=device if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    kern = {}
    for k in ("score_count", "dense", "conv"):
        ms, n = model.profile_read(k)
        kern[k] = (ms / n) if n else None
    ranks_np = ranks.cpu().numpy()

    if rank == 0:
        units = Q * (1 if entity_mode else world) * args.steps
        n_local = model.n_local
        d = int(md["ent_emb_size"])
        out = {
            "metric": "scored triples/sec (1-vs-all)", "value": units / dt, "unit": "triples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if entity_mode else "weak", "vs_baseline": None,
            "dtype": "f32" if args.score_mode == "f32" else "bf16x3 (fp32 values split into two bf16 terms, 3 bf16 MFMAs per product, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "%s: |E|=%d R2=%d d=%d r=%d, Q=%d queries/pass%s, %s relation order" % (
                args.workload, md["num_ent"], md["num_rel"], d, md["rel_emb_size"], Q,
                "" if entity_mode else " per GPU", args.order),
                "parallelism": ("entity-sharded x%d%s" % (world, ", top-%d exchanged" % args.topk if args.topk else ""))
                if entity_mode else ("query-sharded x%d" % world),
                "score_mode": "f32 (v_mfma_f32_32x32x2_f32, exact)" if args.score_mode == "f32" else
                "bf16x3 (3 x v_mfma_f32_32x32x16_bf16 per product, ~2^-16 rel.)", "prepare_ms": round(prepare_ms, 2),
                "mean_rank": float(np.mean(ranks_np)), "mrr": float(np.mean(1.0 / ranks_np))},
        }
        F = model.fc_input_size
        kinfo = {}
        if kern["score_count"]:
            fl = 2.0 * Q * n_local * d           # ALGORITHMIC flops of one score_count launch
            # ALGORITHMIC bytes (fused rank): the entity table once (fp32 rows or two bf16 planes: 4 B per
            # value either way) + pred_bias + h in + counters out
            by = n_local * d * 4.0 + n_local * 4.0 + Q * d * 4.0 + Q * 8.0
            peak = PEAK_F32_MFMA_TFLOPS if args.score_mode == "f32" else PEAK_BF16_MFMA_TFLOPS
            t_ms = kern["score_count"]
            kname = "k_score_count_f32" if args.score_mode == "f32" else "k_score_count_bf16x3"
            # which roof is lower at this shape: few queries against a table larger than the caches -> HBM
            if by / (PEAK_HBM_GBS * 1e9) > fl / (peak * 1e12):
                ach = by / (t_ms * 1e-3) / 1e9
                kinfo[kname] = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                                "avg_launch_ms": t_ms, "algorithmic_bytes": by, "algorithmic_tflops": fl / (t_ms * 1e-3) / 1e12,
                                "traffic": pmc_traffic(args.workload, Q, "coper::" + kname) if world == 1 else None}
            else:
                ach = fl / (t_ms * 1e-3) / 1e12
                kinfo[kname] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                                "avg_launch_ms": t_ms, "algorithmic_gbs": by / (t_ms * 1e-3) / 1e9,
                                "traffic": pmc_traffic(args.workload, Q, "coper::" + kname) if world == 1 else None}
                if args.score_mode != "f32":
                    kinfo[kname]["note"] = "3 hardware bf16 MFMAs per algorithmic product: hardware MFMA utilisation = 3 x frac"
                pm = pmc_mfma_busy(args.workload, Q, "coper::" + kname) if world == 1 else None
                if pm:
                    kinfo[kname]["pmc"] = pm      # counters: fraction of the kernel's cycles with the matrix pipes busy
        if kern["dense"]:
            fl = 2.0 * Q * F * d                 # ALGORITHMIC flops of the dense launch pair (small + big tiles)
            if args.score_mode == "f32":
                ach = fl / (kern["dense"] * 1e-3) / 1e12
                kinfo["k_dense_big_f32"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                            "frac": ach / PEAK_F32_MFMA_TFLOPS, "avg_launch_ms": kern["dense"],
                                            "traffic": pmc_traffic(args.workload, Q, "coper::k_dense_big_f32<13>") if world == 1 else None}
            else:
                # ALGORITHMIC bytes (cached per-relation weights): G weight streams (one per
                # relation tile of <= 128 queries) of F*d values in two bf16 planes + the e1 rows in + h out.
                # The conv runs inside this kernel (x never touches HBM); "dense" times it together with the
                # launch that serves the <= 32-query tiles.
                cnt = np.bincount(q["rel"]) if md.get("context_rel_out", None) is not None else np.array([Q])
                G = int(np.sum((cnt[cnt > 0] + 127) // 128))
                by = G * F * d * 4.0 + Q * d * 4.0 + Q * d * 4.0
                ach = by / (kern["dense"] * 1e-3) / 1e9
                kinfo["k_dense_fused_bf16x3"] = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                                 "frac": ach / PEAK_HBM_GBS, "avg_launch_ms": kern["dense"],
                                                 "algorithmic_bytes": by,
                                                 "algorithmic_tflops": (fl + 2.0 * Q * F * 9) / (kern["dense"] * 1e-3) / 1e12,
                                                 "traffic": pmc_traffic(args.workload, Q, "coper::k_dense_fused_bf16x3") if world == 1 else None}
        if kern["conv"] and args.score_mode == "f32":
            by = Q * (F * 4.0 + d * 4.0)         # ALGORITHMIC bytes: x written (fp32 or two bf16 planes) + e1 row read
            ach = by / (kern["conv"] * 1e-3) / 1e9
            kinfo["k_conv3x3_bn_relu" if args.score_mode == "f32" else "k_conv3x3_bn_relu_bf16"] = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                          "frac": ach / PEAK_HBM_GBS, "avg_launch_ms": kern["conv"],
                                          "traffic": pmc_traffic(args.workload, Q, "coper::k_conv3x3_bn_relu<4>" if args.score_mode == "f32" else "coper::k_conv3x3_bn_relu_bf16<4>") if world == 1 else None}
        if kinfo:
            dom = max(kinfo, key=lambda k: kinfo[k]["avg_launch_ms"])   # the dominant kernel of the step
            out["roofline"] = dict(kernel=dom, **kinfo[dom])
            out["roofline"]["all_kernels"] = {k: {kk: vv for kk, vv in v.items() if kk != "note"} for k, v in kinfo.items() if k != dom}
        if not args.no_cpu_baseline and world == 1:
            host_p = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in params.items()} if not big else None
            if host_p is not None:
                out["cpu_baseline"] = cpu_baseline(md, host_p, q, args.cpu_seconds)
                out["config"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(out))
        sys.stdout.flush()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
