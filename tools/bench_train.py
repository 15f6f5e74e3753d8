#!/usr/bin/env python3
"""Training-step timing: `python tools/bench_train.py [workload] [B] [L]`.
One step = one batch of B (e1, rel) queries scored against L sampled entities: forward, backward, clip, AMSGrad."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from coper_amd import data as cdata
from coper_amd.models import ConvE
name = sys.argv[1] if len(sys.argv) > 1 else "fb15k237_cpg"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
L = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
md = cdata.model_descriptors(name)
md.update(batch_norm_train_stats=True, batch_norm_momentum=0.1, hidden_dropout=0.3, output_dropout=0.2,
          label_smoothing_epsilon=0.1, learning_rate=1e-3)
p = cdata.synthetic_params(md, 0)
m = ConvE(md, device="cuda:0").load_parameters(p)
m.train_init(seed=1)
rng = np.random.default_rng(0)
def batch():
    lk = rng.integers(0, md["num_ent"], (B, L)).astype(np.int32)
    lab = (rng.random((B, L)) < 0.01).astype(np.float32)
    return dict(e1=torch.as_tensor(rng.integers(0, md["num_ent"], B)).cuda(), rel=torch.as_tensor(rng.integers(0, md["num_rel"], B)).cuda(),
                lookup_values=torch.as_tensor(lk).cuda(), e2_multi=torch.as_tensor(lab).cuda())
bs = [batch() for _ in range(4)]
for i in range(3):
    loss = m.train_step(bs[i % 4])
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for i in range(K):
    loss = m.train_step(bs[i % 4])
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / K
# Roofline of the step: ALGORITHMIC work = what any implementation of models.py:186-200 + amsgrad.py:130-189 must do.
#   bytes: AMSGrad reads p, g, m, v, v_hat and writes p, m, v, v_hat of EVERY trainable element each step (dense decay of all slots,
#          amsgrad.py:141-159): 9 x 4 B per parameter; + the sampled rows gathered twice (forward, backward) B L d 4 B each;
#   flops: generated dense layer: three products of 2 B r F d (T = x P, dP, dx; static layer: 2 B F d each), the scorer 2 B L d x 3.
n_param = sum(int(np.prod(v.shape)) for k, v in m._tensors.items() if not k.endswith(("moving_mean", "moving_variance")))
dm = cdata._dims(md)
F, d, r = dm["F"], int(md["ent_emb_size"]), int(md["rel_emb_size"])
gen = md.get("context_rel_out", None) is not None and not md.get("do_parameter_lookup", False)
by = 9.0 * 4.0 * n_param + 2.0 * B * L * d * 4.0
fl = 3.0 * 2.0 * B * (r if gen else 1) * F * d + 3.0 * 2.0 * B * L * d
t_hbm, t_mfma = by / 8.0e12, fl / 2.5e15
floor_ms = max(t_hbm, t_mfma) * 1e3
roof = {"bound": "hbm" if t_hbm >= t_mfma else "mfma", "algorithmic_bytes": by, "algorithmic_flops": fl, "floor_ms": floor_ms,
        "frac": floor_ms / ms, "achieved": (by / (ms * 1e-3) / 1e9) if t_hbm >= t_mfma else (fl / (ms * 1e-3) / 1e12),
        "peak": 8000.0 if t_hbm >= t_mfma else 2500.0, "unit": "GB/s" if t_hbm >= t_mfma else "TFLOP/s",
        "note": "whole step against the larger of its two floors (AMSGrad's dense slot traffic at 8 TB/s; the three dense-layer products "
                "and the scorer at 2.5 PF -- the x3 arithmetic spends three hardware MFMAs per product); host clock over %d steps" % K}
print(json.dumps({"metric": "training step", "workload": name, "B": B, "L": L, "ms_per_step": ms, "queries_per_s": B / ms * 1e3,
                  "scored_pairs_per_s": B * L / ms * 1e3, "loss": float(loss.cpu()[0]), "dtype": "fp16x3", "trainable_parameters": n_param,
                  "roofline": roof}))

# the same step fed by the samplers: a synthetic train graph of this shape, one known tail list per (e1, rel)
if "--with-sampler" in sys.argv:
    from coper_amd.data import DeviceTrainDataset
    N = 100000
    ip = np.zeros(N + 1, np.int64); ip[1:] = np.cumsum(rng.integers(1, 4, N))
    s = dict(e1=rng.integers(0, md["num_ent"], N), rel=rng.integers(0, md["num_rel"], N), tail_indptr=ip,
             tail_idx=rng.integers(0, md["num_ent"], ip[-1]))
    for one_pos, prop in ((True, 10.0), (False, 100.0)):      # (False, 100): config_FB15k-237_cpg.yaml:21-23
        it = iter(DeviceTrainDataset(s, md["num_ent"], B, num_labels=L, device="cuda:0", one_positive_label_per_sample=one_pos,
                                     prop_negatives=prop))
        for _ in range(3):
            m.train_step(next(it))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K):
            loss = m.train_step(next(it))
        torch.cuda.synchronize()
        ms2 = (time.perf_counter() - t0) * 1e3 / K
        print(json.dumps({"metric": "training step fed by DeviceTrainDataset", "sampler": "one positive per row" if one_pos else
                          "proportional (prop_negatives %g)" % prop, "workload": name, "B": B, "L": L, "ms_per_step": ms2,
                          "queries_per_s": B / ms2 * 1e3}))
