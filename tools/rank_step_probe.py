"""ONE rank's compute of an 8-rank pass of the 10M-entity config (bench.py's scale.projected), alone in a process so that a
rocprofv3 --kernel-trace --stats run shows where its time goes:  rocprofv3 --kernel-trace --stats -d out -- python3 tools/rank_step_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from coper_amd import data as cdata
from coper_amd.models import ConvE
from coper_amd.sharding import shard_bounds

G = int(os.environ.get("G", "8")); K = int(os.environ.get("K", "10"))
dev = torch.device("cuda:0")
md = cdata.model_descriptors("synth10m_cpg")
Q, d = 4096, int(md["ent_emb_size"])
q = cdata.synthetic_queries(md, Q, seed=0)
shard = shard_bounds(md["num_ent"], G, 0)
params, _ = bench.device_params(md, 0, dev, shard)
m = ConvE(md, device=dev, shard=shard, score_mode="bf16x3")
m.load_parameters(params, global_rows=False); m.prepare(); m.reserve(Q, len(q["filt_idx"]))
dq = {n: torch.as_tensor(v).to(dev) for n, v in q.items()}
mine = np.nonzero(q["rel"] % G == 0)[0]
sel = torch.as_tensor(mine, device=dev)
rows1 = torch.randn((Q, d), device=dev) * 0.1; rows2 = torch.randn((Q, d), device=dev) * 0.1
bias2 = torch.zeros(Q, device=dev); hfull = torch.randn((Q, d), device=dev).abs()
def step(k=K):
    ts = [time.perf_counter()]
    def mark():
        if os.environ.get("SYNC"):
            torch.cuda.synchronize(); ts.append(time.perf_counter())
    r1, r2, b2 = m.gather_entities(dq["e1"]), m.gather_entities(dq["e2"]), m.gather_bias(dq["e2"]); mark()
    hloc = m.encode(q["e1"][mine], q["rel"][mine], e1_rows=rows1.index_select(0, sel).contiguous()); mark()
    hfull.index_copy_(0, sel, hloc)
    tx = m.score_rows(hfull, rows2, bias2); mark()
    out = m.rank_counts(hfull, torch.stack([tx, tx]), dq["e2"], dq["filt_indptr"], dq["filt_idx"], filt_nnz=len(q["filt_idx"]), k=k); mark()
    return ts
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print("rank step (k=%d): %.3f ms" % (K, (time.perf_counter() - t0) * 100))
os.environ["SYNC"] = "1"
acc = np.zeros(4)
for _ in range(5):
    ts = step(); acc += np.diff(ts)
print("with a synchronize after each part: gathers %.3f  encode %.3f  targets %.3f  rank_counts %.3f ms" % tuple(acc / 5 * 1e3))
del os.environ["SYNC"]
for k in (0,):
    for _ in range(2): step(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step(k)
    torch.cuda.synchronize(); print("rank step (k=%d): %.3f ms" % (k, (time.perf_counter() - t0) * 100))
