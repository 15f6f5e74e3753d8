"""PyTorch-CPU restatement of the evaluation pass: the `cpu_baseline` of SURVEY.md 8(d).  TEST INFRASTRUCTURE ONLY.

Only `bench.py`'s `cpu_baseline` leg and `tests/` may import this module (nothing under `coper_amd/` does).

The reference's CPU path is TensorFlow-1.14 CPU kernels + a NumPy ranker; TensorFlow cannot run here
(SURVEY.md 8(c)).  This module runs the same graph, op for op, on the same class of library kernels
(oneDNN / MKL through torch on the host cores), fp32:

* `models.py:176-180`  embedding_lookup of e1 / rel           -> index_select
* `models.py:355-362`  reshape, stack (plain ConvE)           -> view / cat
* `models.py:372-391`  conv2d VALID (static filters) or per-sample filters (`tf.map_fn`) + bias, BN, ReLU
                                                              -> F.conv2d (groups = B for per-sample filters)
* `models.py:56-76`    generator: chain of bias-free matmuls  -> mm
* `models.py:70,350`   generated dense weights materialised as a [B, F, d] tensor
* `models.py:412`      batched [B,1,F] x [B,F,d]              -> bmm
* `models.py:416-419`  BN, ReLU
* `models.py:434-437`  h . ent_emb^T + pred_bias              -> addmm
* `metrics.py:44-57`   per row: mask known answers to -inf, restore the target, full argsort, position of e2
                       -- NumPy, literally as the reference does it.

`tests/test_oracle_golden.py` checks it against `oracle/coper_oracle.py` (itself pinned to the reference) on the
golden forward fixtures."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .coper_oracle import BN_EPS, Dims

__all__ = ["TorchCPUModel"]


def _fold_bn(p, prefix):
    g, b = p[prefix + "/gamma"], p[prefix + "/beta"]
    m, v = p[prefix + "/moving_mean"], p[prefix + "/moving_variance"]
    inv = g / torch.sqrt(v + BN_EPS)
    return inv, b - m * inv


class TorchCPUModel(object):
    """Weights as CPU fp32 tensors by the reference's leaf names; `eval_pass` = one evaluation pass the way
    `run_cpg.py:18-35 -> metrics.py:38-60` runs it."""

    def __init__(self, params, md, device="cpu", dtype=torch.float32):
        """`device` / `dtype` other than the CPU-baseline defaults are for diagnostics (tools/rank_decomp.py runs the
        same graph in float64 on the GPU to have reference-semantics ranks of a full-size pass in seconds)."""
        self.md, self.dims = md, Dims(md)
        self.device = torch.device(device)
        self.p = {k: torch.as_tensor(np.ascontiguousarray(np.asarray(v, dtype=np.float32))).to(device=self.device, dtype=dtype)
                  for k, v in params.items()}

    def _generate(self, ctx, name, hidden):
        p, v = self.p, ctx
        for i in range(len(hidden)):
            v = v @ p["%s/CPG/Projection%d" % (name, i)]
            if self.dims.ctx_bn:
                sc, sh = _fold_bn(p, "%s/CPG/Projection%d/BatchNorm" % (name, i))
                v = v * sc + sh
            v = torch.relu(v)
        return v @ p["%s/CPG/Projection%d" % (name, len(hidden))]

    def _param(self, name, ctx, rel, hidden):
        if self.dims.lookup:
            return self.p[name].index_select(0, rel)
        return self._generate(ctx, name, hidden)

    @torch.no_grad()
    def forward(self, e1, rel):
        dm, p = self.dims, self.p
        e1 = torch.as_tensor(np.asarray(e1, dtype=np.int64)).to(self.device)
        rel = torch.as_tensor(np.asarray(rel, dtype=np.int64)).to(self.device)
        B = e1.shape[0]
        x0 = p["ent_emb"].index_select(0, e1)
        c = None if dm.lookup else p["rel_emb"].index_select(0, rel)
        img = x0.view(B, 1, dm.emb_h, dm.emb_w)
        if dm.stacked:
            img = torch.cat([img, c.view(B, 1, dm.emb_h, dm.r // dm.emb_h)], dim=2)
        if dm.gen_conv:   # per-sample filters (tf.map_fn over the batch, models.py:375-380): one grouped convolution
            K = self._param("conv1_weights", c, rel, dm.ctx_conv).view(B, dm.fh, dm.fw, dm.C)
            kb = self._param("conv1_bias", c, rel, dm.ctx_conv).view(B * dm.C)
            w = K.permute(0, 3, 1, 2).reshape(B * dm.C, 1, dm.fh, dm.fw)
            y = F.conv2d(img.view(1, B, dm.in_h, dm.in_w), w, kb, groups=B).view(B, dm.C, dm.Ho, dm.Wo)
        else:
            w = p["conv1_weights"].view(dm.fh, dm.fw, dm.C).permute(2, 0, 1).unsqueeze(1).contiguous()   # HWIO -> OIHW
            y = F.conv2d(img, w, p["conv1_bias"])
        sc, sh = _fold_bn(p, "Conv1BN")
        y = torch.relu(y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
        x = y.permute(0, 2, 3, 1).reshape(B, -1)          # NHWC flatten order (models.py:404)
        if dm.concat_rel:
            x = torch.cat([x, c], dim=1)
        if dm.gen_fc:
            W = self._param("fc_weights", c, rel, dm.ctx_out).view(B, dm.F, dm.d)        # [B,F,d] materialised
            bvec = self._param("fc_bias", c, rel, dm.ctx_out).view(B, dm.d)
            z = torch.bmm(x.unsqueeze(1), W).squeeze(1) + bvec
        else:
            z = torch.addmm(p["fc_bias"], x, p["fc_weights"])
        sc, sh = _fold_bn(p, "FCBN")
        return torch.relu(z * sc + sh)

    @torch.no_grad()
    def predictions_all(self, h):
        return torch.addmm(self.p["pred_bias"], h, self.p["ent_emb"].t())

    def eval_pass(self, e1, rel, e2, indptr, idx, batch_size=512):
        """Ranks int64 [Q]: forward + logits on torch-CPU, then the reference's ranker loop in NumPy."""
        E = self.dims.num_ent
        ranks = []
        for s in range(0, len(e1), batch_size):
            t = min(len(e1), s + batch_size)
            pred = self.predictions_all(self.forward(e1[s:t], rel[s:t])).numpy()
            e2b = np.asarray(e2[s:t])
            e2_multi = np.zeros((t - s, E), dtype=np.float32)            # data.py:182-186: sparse_to_dense
            for i in range(t - s):
                e2_multi[i, idx[indptr[s + i]:indptr[s + i + 1]]] = 1.0
            for i in range(t - s):                                       # metrics.py:43-50
                target_value = pred[i, e2b[i]]
                pred[i][e2_multi[i] == 1] = -np.inf
                pred[i, e2b[i]] = target_value
                argsort = np.argsort(-pred[i])
                ranks.append(int(np.where(argsort == e2b[i])[0][0]) + 1)
        return np.asarray(ranks, dtype=np.int64)
