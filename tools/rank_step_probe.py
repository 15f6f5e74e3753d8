"""ONE rank's work of an 8-rank evaluation of the 10M-entity config (bench.py's scale.projected), alone in a process so that a
rocprofv3 --kernel-trace --stats run shows where its time goes:
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/rank_step_probe.py          (G=8 K=10 OVERLAP=1 in the environment)
Round 6: the rank runs through EntityShardedRanker.rank_stream itself (emulate_world: plans rebuilt per chunk, every launch of the
exchange, one handle per role, steps 1 - 2 of the next chunk on the side stream), the all-gathers replaced by local copies."""
import itertools, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from coper_amd import data as cdata
from coper_amd.models import ConvE
from coper_amd.sharding import EntityShardedRanker, shard_bounds

G = int(os.environ.get("G", "8")); K = int(os.environ.get("K", "10")); OVERLAP = os.environ.get("OVERLAP", "1") != "0"
N = int(os.environ.get("CHUNKS", "12"))
dev = torch.device("cuda:0")
md = cdata.model_descriptors("synth10m_cpg")
Q = 4096
qs = [cdata.synthetic_queries(md, Q, seed=s) for s in range(4)]
shard = shard_bounds(md["num_ent"], G, 0)
params, _ = bench.device_params(md, 0, dev, shard)
ms = ConvE(md, device=dev, shard=shard, score_mode="bf16x3", role="score").load_parameters(params, global_rows=False)
me = ConvE(md, device=dev, shard=shard, score_mode="bf16x3", role="encode", rel_mod=(G, 0)).load_parameters(params, global_rows=False)
er = EntityShardedRanker(ms, encoder=me, emulate_world=(G, 0), overlap=OVERLAP)
ms.prepare(); me.prepare()
ms.reserve(Q, max(len(q["filt_idx"]) for q in qs)); me.reserve(Q, 0)
chunks = []
for q in qs:
    dq = {n: torch.as_tensor(q[n]).to(dev) for n in ("e2", "filt_indptr", "filt_idx")}
    chunks.append(dict(e1=q["e1"], rel=q["rel"], e2=q["e2"], e2_dev=dq["e2"], filt_indptr=dq["filt_indptr"], filt_idx=dq["filt_idx"]))
list(er.rank_stream(chunks, k=K, window=4))
torch.cuda.synchronize(); t0 = time.perf_counter()
list(er.rank_stream(itertools.islice(itertools.cycle(chunks), N), k=K, window=4))
torch.cuda.synchronize()
print("one rank of %d, %d chunks of %d queries through rank_stream (overlap %s, top-%d): %.3f ms per chunk" % (G, N, Q, er.overlap, K, (time.perf_counter() - t0) / N * 1e3))
# the host's share alone: the plans
t0 = time.perf_counter()
for c in itertools.islice(itertools.cycle(chunks), 40):
    er.plan(c)
print("host plan of a chunk: %.3f ms" % ((time.perf_counter() - t0) / 40 * 1e3))
