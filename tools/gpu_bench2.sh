cd $GRAFT_REPO_ROOT
for st in 20 100 400; do timeout 300 python bench.py --no-cpu-baseline --no-scale --no-extras --steps $st --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $st: value %.3fM ms/step %.4f' % (d['value']/1e6, d['ms_per_step']))"; done
python3 - <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from coper_amd import data as cdata
from coper_amd.models import ConvE
md = cdata.model_descriptors("fb15k237_cpg"); p = cdata.synthetic_params(md, 0)
m = ConvE(md, device="cuda:0", score_mode="bf16x3").load_parameters(p).prepare()
q = cdata.synthetic_queries(md, 20480, seed=0); dq = {k: torch.as_tensor(v).cuda() for k, v in q.items()}
nnz = len(q["filt_idx"]); m.reserve(20480, nnz)
def step(): return m.rank_pass(dq["e1"], dq["rel"], dq["e2"], dq["filt_indptr"], dq["filt_idx"], filt_nnz=nnz, want_equal=False)
for _ in range(5): step()
torch.cuda.synchronize()
# host time to enqueue one pass
t0 = time.perf_counter()
for _ in range(20): step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print("20 passes: host enqueue %.3f ms/pass, wall incl. sync %.3f ms/pass" % (t_enq / 20 * 1e3, t_all / 20 * 1e3))
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for a, b in ev:
    a.record(); step(); b.record()
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
ts = [a.elapsed_time(b) for a, b in ev]
print("per-pass GPU ms:", " ".join("%.3f" % t for t in ts), "| wall %.3f ms/pass, first-start to last-end %.3f" % (t_all / 20 * 1e3, ev[0][0].elapsed_time(ev[-1][1]) / 20))
PY
