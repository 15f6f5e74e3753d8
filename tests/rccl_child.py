"""Child process of tests/test_gpu_rccl.py: creates a REAL RCCL communicator (backend "nccl", one rank -- the box has one
GPU) and runs both sharded evaluators through it: the all-reduces of the entity-sharded exchange and the packed
all-gather (counts + top-k) execute as RCCL collectives on the device.  Results must equal the single-process ones."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29611")
    os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"
    import numpy as np
    import torch
    import torch.distributed as dist
    from coper_amd import data as cdata
    from coper_amd.models import ConvE
    from coper_amd.sharding import EntityShardedRanker, QueryShardedEvaluator

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    md = cdata.model_descriptors("fb15k237_cpg", num_ent=5003, num_rel=40)
    p = cdata.synthetic_params(md, 5)
    q = cdata.synthetic_queries(md, 700, seed=6)
    for mode in ("bf16x3", "f32"):
        m = ConvE(md, device=dev, score_mode=mode).load_parameters(p).prepare()
        # single-process results first (no process group yet in the first round; compared by value afterwards)
        r0, ne0 = m.rank_pass(q["e1"], q["rel"], q["e2"], q["filt_indptr"], q["filt_idx"])
        h = m.encode(q["e1"], q["rel"])
        tgt = m.target_scores(h, q["e2"])
        _, _, tv0, ti0 = m.rank_counts(h, tgt, q["e2"], q["filt_indptr"], q["filt_idx"], k=10)
        if not dist.is_initialized():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        assert dist.get_backend() == "nccl"
        er = EntityShardedRanker(m)
        assert er.dist and er.world == 1
        ranks, ne, tv, ti = er.rank(q, k=10)            # 3 all-reduces + the packed all-gather, all on RCCL
        assert torch.equal(ranks, r0) and torch.equal(ne, ne0), mode
        assert torch.equal(tv, tv0) and torch.equal(ti, ti0), mode
        ranks2, ne2 = er.rank(q, k=0)
        assert torch.equal(ranks2, r0) and torch.equal(ne2, ne0)
        # round 6: the evaluation loop with one handle per role (coper_config.role) -- steps 1 - 2 of chunk n + 1 on a side stream and
        # a second RCCL communicator under chunk n's count launch; three different chunks, window of two, the first chunk's results
        qs = [q] + [cdata.synthetic_queries(md, 500 + 40 * i, seed=30 + i) for i in range(2)]
        want = [er.rank(c, k=10) for c in qs]
        ms = ConvE(md, device=dev, score_mode=mode, role="score").load_parameters(p).prepare()
        me = ConvE(md, device=dev, score_mode=mode, role="encode").load_parameters(p).prepare()
        er2 = EntityShardedRanker(ms, encoder=me, overlap=True, side_communicator=True)      # (default: off with one rank)
        assert er2.overlap and er2.side is not None
        for rep in range(2):
            got = list(er2.rank_stream(iter(qs), k=10, window=2))
            for a, b in zip(got, want):
                for x, y in zip(a, b):
                    assert torch.equal(x, y), (mode, rep)
        ms.close(); me.close()
        qe = QueryShardedEvaluator(m)
        r3, ne3 = qe.rank(q)                            # all-gather of the int32 ranks
        assert torch.equal(r3, r0) and torch.equal(ne3, ne0), mode
        # a plain collective on the communicator, checked by value
        t = torch.arange(8, device=dev, dtype=torch.float32)
        dist.all_reduce(t)
        assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
        m.close()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RCCL_CHILD_OK backend=nccl world=1 mean_rank=%.3f" % float(np.mean(r0.cpu().numpy())))


if __name__ == "__main__":
    main()
