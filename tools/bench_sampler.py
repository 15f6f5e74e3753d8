#!/usr/bin/env python3
"""The device samplers alone: `python tools/bench_sampler.py [workload] [B] [L]` -- ms per batch of `DeviceTrainDataset`
(one positive per row / proportional), nothing consuming the batches, on a synthetic train graph of the workload's shape."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from coper_amd import data as cdata
from coper_amd.data import DeviceTrainDataset
name = sys.argv[1] if len(sys.argv) > 1 else "fb15k237_cpg"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
L = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
md = cdata.model_descriptors(name)
rng = np.random.default_rng(0)
N = 100000
ip = np.zeros(N + 1, np.int64); ip[1:] = np.cumsum(rng.integers(1, 4, N))
s = dict(e1=rng.integers(0, md["num_ent"], N), rel=rng.integers(0, md["num_rel"], N), tail_indptr=ip, tail_idx=rng.integers(0, md["num_ent"], ip[-1]))
for one_pos, prop in ((True, 10.0), (False, 100.0)):
    it = iter(DeviceTrainDataset(s, md["num_ent"], B, num_labels=L, device="cuda:0", one_positive_label_per_sample=one_pos, prop_negatives=prop))
    for _ in range(5):
        next(it)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 50
    for _ in range(K):
        b = next(it)
    t_host = (time.perf_counter() - t0) * 1e3 / K
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) * 1e3 / K
    print(json.dumps({"sampler": "one positive per row" if one_pos else "proportional (prop_negatives %g)" % prop, "workload": name, "B": B, "L": L,
                      "ms_per_batch_host": t_host, "ms_per_batch": t_all}))
