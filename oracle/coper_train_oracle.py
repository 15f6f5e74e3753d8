"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's TRAINING step, used by
tests/ as the checker for `coper_train_step`.  Never imported by the product.

Parity status: **unpinned** against TensorFlow (no TF in this environment, and the reference has no
training tests or golden vectors); the forward restates `CoPER_ConvE/qa_cpg/models.py` line by line
and the gradients come from torch autograd over that restatement in float64, so what the tests pin
is "the HIP step == the derivative of the documented forward", plus the optimizer arithmetic of
`qa_cpg/utils/amsgrad.py`.  PINNED to reference code for the structure the PyTorch sister models share
(conv -> dense, static or g_linear / g_MLP(+BN) generated -> FCBN -> 1-vs-all scorer -> label-smoothed BCE, BN on moving
statistics, no dropout): tests/golden/minerva_grads.npz holds loss and gradients from torch autograd through
the reference's own `fact_network.py` forward and `emb.py:50-58` loss; this oracle (test_train_oracle.py) and the
HIP step (test_gpu_train.py) both reproduce them.

What is restated (reference file:line):
  * train-mode forward: gather, reshape [B,H,W,1], (plain: stack rel image, models.py:360-362),
    VALID 3x3 cross-correlation + bias (:382-385), Conv1BN with batch statistics when
    `batch_norm_train_stats` (:358,:386-388), ReLU, dropout(hidden_dropout) (:389-391), NHWC flatten
    (:404), static or generated dense (:410-412; generator = bias-free projection chain, :56-76),
    dropout(output_dropout) THEN FCBN THEN ReLU (:414-419), sampled scorer (:438-443).
  * loss: t' = (1 - eps)*t + 1/num_ent, mean over B*L of sigmoid cross-entropy with logits (:448-453).
  * BN moving statistics: moving = moving*momentum + batch*(1 - momentum)  (momentum is the decay,
    SURVEY.md section 9); [TF-semantics] Conv1BN (4-D, fused kernel) feeds the UNBIASED batch
    variance into the moving average, FCBN (2-D, unfused fallback) the biased one; both normalise
    with the biased variance, epsilon 1e-3.
  * clip_by_global_norm(5.0) (:199) and AMSGrad (amsgrad.py:130-159; sparse == dense with zero rows:
    every slot decays every step, :166-189): lr_t = lr*sqrt(1 - b2^t)/(1 - b1^t); m = b1*m + (1-b1)*g;
    v = b2*v + (1-b2)*g*g; v_hat = max(v_hat, v); p -= lr_t*m/(sqrt(v_hat) + eps).
  * dropout masks: TF's RNG cannot be reproduced, so the masks are an input: `dropout_keep()` below is the
    counter-based hash the HIP kernels use (same bits), keep-scaled by 1/(1 - rate) as tf.nn.dropout.
"""
from __future__ import annotations

import numpy as np
import torch

BN_EPS = 1e-3


def dropout_keep(seed: int, step: int, stage: int, n: int, rate: float) -> np.ndarray:
    """0/1 keep mask of n elements; bit-identical to `dropout_keep_u32` in csrc/train_common.h."""
    if rate <= 0.0:
        return np.ones(n, np.float32)
    idx = np.arange(n, dtype=np.uint64)
    x = (idx * np.uint64(0x9E3779B1) + np.uint64(seed) * np.uint64(0x85EBCA77) + np.uint64(step) * np.uint64(0xC2B2AE3D)
         + np.uint64(stage) * np.uint64(0x27D4EB2F)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(12)
    x = (x * np.uint64(0x297A2D39)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    thr = np.uint64(int(np.float32(rate) * np.float32(16777216.0)))   # the float arithmetic of dropout_threshold24()
    return ((x >> np.uint64(8)) >= thr).astype(np.float32)


def _generate(p, md, name, c, train_stats, keep_ctx, stats, dtype):
    """ContextualParameterGenerator.generate (models.py:56-70) in train mode; keep_ctx[(name, i)] = 0/1 dropout mask."""
    ctx = list(md["context_rel_conv"] if name.startswith("conv1") else md["context_rel_out"])
    use_bn = bool(md.get("context_rel_use_batch_norm", False))
    rate = float(md.get("context_rel_dropout", 0.0))
    v = c
    for i in range(len(ctx)):
        v = v @ p["%s/CPG/Projection%d" % (name, i)]
        if use_bn:
            bn = "%s/CPG/Projection%d/BatchNorm" % (name, i)
            if train_stats:
                m = v.mean(dim=0)
                var = ((v - m) ** 2).mean(dim=0)
                stats[bn] = (m.detach(), var.detach(), v.shape[0])
            else:
                m, var = p[bn + "/moving_mean"], p[bn + "/moving_variance"]
            v = (v - m) / torch.sqrt(var + BN_EPS) * p[bn + "/gamma"] + p[bn + "/beta"]
        v = torch.relu(v)
        v = v * torch.as_tensor(keep_ctx[(name, i)].reshape(v.shape)).to(dtype) / (1.0 - rate)
    return v @ p["%s/CPG/Projection%d" % (name, len(ctx))]


def forward_train(p, md, batch, keep_hidden, keep_out, dtype=torch.float64, keep_ctx=None):
    """p: dict leaf name -> torch tensor (requires_grad where trainable).  batch: e1, rel int64 [B],
    lookup int64 [B,L], labels float [B,L].  keep_*: 0/1 masks (numpy) for the two dropouts.
    Returns loss (scalar tensor) and the BN batch statistics dict."""
    d, r = int(md["ent_emb_size"]), int(md["rel_emb_size"])
    H = int(md.get("emb_h", 10))
    Wd = d // H
    C = int(md.get("conv_num_channels", 32))
    ctx_conv, ctx_out = md.get("context_rel_conv", None), md.get("context_rel_out", None)
    lookup_params = bool(md.get("do_parameter_lookup", False))
    if lookup_params and ((ctx_out is None and ctx_conv is None) or md.get("concat_rel", False)):
        raise NotImplementedError("training oracle: g_lookup needs a looked-up layer and cannot concat_rel (models.py:180,263-264,406)")
    train_stats = bool(md.get("batch_norm_train_stats", False))
    e1 = torch.as_tensor(batch["e1"]).long()
    rel = torch.as_tensor(batch["rel"]).long()
    lookup = None if batch.get("lookup", None) is None else torch.as_tensor(batch["lookup"]).long()
    labels = torch.as_tensor(batch["labels"]).to(dtype)
    B = e1.shape[0]
    x0 = p["ent_emb"][e1]
    c = None if lookup_params else p["rel_emb"][rel]          # g_lookup passes the ids through (models.py:180)
    img = x0.reshape(B, H, Wd)
    if ctx_out is None and ctx_conv is None:                # plain ConvE: stack the relation image (models.py:360-362)
        img = torch.cat([img, c.reshape(B, H, r // H)], dim=1)
    stats = {}
    fh, fw = int(md.get("conv_filter_height", 3)), int(md.get("conv_filter_width", 3))   # models.py:109-110
    Ho, Wo = img.shape[1] - fh + 1, img.shape[2] - fw + 1                                # VALID, stride 1 (models.py:382-385)
    y = torch.zeros((B, Ho, Wo, C), dtype=dtype)
    if ctx_conv is None:
        K = p["conv1_weights"].reshape(fh, fw, C)
        for u in range(fh):
            for v in range(fw):
                y = y + img[:, u:u + Ho, v:v + Wo, None] * K[u, v][None, None, None, :]
        y = y + p["conv1_bias"]
    else:                                                   # per-sample filters (models.py:374-380)
        if lookup_params:
            Kb_ = p["conv1_weights"][rel].reshape(B, fh, fw, C)
            kb_ = p["conv1_bias"][rel]
        else:
            Kb_ = _generate(p, md, "conv1_weights", c, train_stats, keep_ctx, stats, dtype).reshape(B, fh, fw, C)
            kb_ = _generate(p, md, "conv1_bias", c, train_stats, keep_ctx, stats, dtype)
        for u in range(fh):
            for v in range(fw):
                y = y + img[:, u:u + Ho, v:v + Wo, None] * Kb_[:, u, v][:, None, None, :]
        y = y + kb_[:, None, None, :]
    if train_stats:
        m1 = y.mean(dim=(0, 1, 2))
        v1 = ((y - m1) ** 2).mean(dim=(0, 1, 2))
        n1 = B * Ho * Wo
        stats["Conv1BN"] = (m1.detach(), v1.detach(), n1)
    else:
        m1, v1 = p["Conv1BN/moving_mean"], p["Conv1BN/moving_variance"]
    y = (y - m1) / torch.sqrt(v1 + BN_EPS) * p["Conv1BN/gamma"] + p["Conv1BN/beta"]
    y = torch.relu(y)
    hd = float(md.get("hidden_dropout", 0.0))
    x = y.reshape(B, -1)
    x = x * torch.as_tensor(keep_hidden.reshape(B, -1)).to(dtype) / (1.0 - hd)
    if md.get("concat_rel", False):                         # models.py:406-407 (after the dropout)
        x = torch.cat([x, c], dim=1)
    F = x.shape[1]
    if ctx_out is None:
        z = x @ p["fc_weights"] + p["fc_bias"]
    elif lookup_params:
        Wg = p["fc_weights"][rel].reshape(B, F, d)                       # ParameterLookup.generate (models.py:90-94)
        z = torch.einsum("bf,bfk->bk", x, Wg) + p["fc_bias"][rel]
    else:
        Wg = _generate(p, md, "fc_weights", c, train_stats, keep_ctx, stats, dtype).reshape(B, F, d)   # models.py:70,73
        bg = _generate(p, md, "fc_bias", c, train_stats, keep_ctx, stats, dtype)
        z = torch.einsum("bf,bfk->bk", x, Wg) + bg                       # models.py:412
    od = float(md.get("output_dropout", 0.0))
    z = z * torch.as_tensor(keep_out.reshape(B, -1)).to(dtype) / (1.0 - od)
    if train_stats:
        m2 = z.mean(dim=0)
        v2 = ((z - m2) ** 2).mean(dim=0)
        stats["FCBN"] = (m2.detach(), v2.detach(), B)
    else:
        m2, v2 = p["FCBN/moving_mean"], p["FCBN/moving_variance"]
    z = (z - m2) / torch.sqrt(v2 + BN_EPS) * p["FCBN/gamma"] + p["FCBN/beta"]
    h = torch.relu(z)
    if lookup is None:
        s = h @ p["ent_emb"].T + p["pred_bias"]                                        # models.py:434-437
    else:
        s = torch.einsum("bk,blk->bl", h, p["ent_emb"][lookup]) + p["pred_bias"][lookup]   # models.py:439-443
    eps_ls = float(md.get("label_smoothing_epsilon", 0.0))
    t = (1.0 - eps_ls) * labels + 1.0 / float(md["num_ent"])               # models.py:450 (1/|E|, not eps/|E|)
    per = torch.clamp(s, min=0) - s * t + torch.log1p(torch.exp(-torch.abs(s)))
    loss = per.mean()
    return loss, stats, h, s


TRAINABLE_STATIC = ["ent_emb", "rel_emb", "conv1_weights", "conv1_bias", "pred_bias", "Conv1BN/gamma", "Conv1BN/beta",
                    "FCBN/gamma", "FCBN/beta"]


def trainable_names(md):
    names = list(TRAINABLE_STATIC)
    gen_conv = md.get("context_rel_conv", None) is not None
    if md.get("do_parameter_lookup", False):
        names.remove("rel_emb")
        names += ["fc_weights", "fc_bias"]            # tables; conv1_weights / conv1_bias are tables too when looked up
        return names
    gens = []
    if md.get("context_rel_out", None) is None:
        names += ["fc_weights", "fc_bias"]
    else:
        gens += [("fc_weights", md["context_rel_out"]), ("fc_bias", md["context_rel_out"])]
    if gen_conv:
        names.remove("conv1_weights")
        names.remove("conv1_bias")
        gens += [("conv1_weights", md["context_rel_conv"]), ("conv1_bias", md["context_rel_conv"])]
    for g, ctx in gens:
        nh = len(ctx)
        for i in range(nh + 1):
            names.append("%s/CPG/Projection%d" % (g, i))
            if i < nh and md.get("context_rel_use_batch_norm", False):
                names += ["%s/CPG/Projection%d/BatchNorm/gamma" % (g, i), "%s/CPG/Projection%d/BatchNorm/beta" % (g, i)]
    return names


class AMSGrad(object):
    """amsgrad.py:130-159 (dense form; the sparse form is the same arithmetic with zero rows)."""

    def __init__(self, names, params, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, clip=5.0):
        self.lr, self.b1, self.b2, self.eps, self.clip = lr, beta1, beta2, eps, clip
        self.b1p, self.b2p = beta1, beta2                        # beta powers start at beta (amsgrad.py:108-113)
        self.m = {k: np.zeros_like(params[k], dtype=np.float64) for k in names}
        self.v = {k: np.zeros_like(params[k], dtype=np.float64) for k in names}
        self.vh = {k: np.zeros_like(params[k], dtype=np.float64) for k in names}

    def step(self, params, grads):
        gn = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values()))
        scale = self.clip / max(gn, self.clip)                   # tf.clip_by_global_norm
        lr_t = self.lr * np.sqrt(1.0 - self.b2p) / (1.0 - self.b1p)
        for k, g in grads.items():
            g = g.astype(np.float64) * scale
            self.m[k] = self.b1 * self.m[k] + (1.0 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1.0 - self.b2) * g * g
            self.vh[k] = np.maximum(self.vh[k], self.v[k])
            params[k] = params[k] - lr_t * self.m[k] / (np.sqrt(self.vh[k]) + self.eps)
        self.b1p *= self.b1
        self.b2p *= self.b2
        return gn


def train_step(params_np, md, batch, opt: AMSGrad, seed, step, momentum):
    """One reference-semantics step in float64.  Mutates params_np (incl. BN moving statistics).  Returns
    (loss, grads dict, global grad norm)."""
    names = trainable_names(md)
    p = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=(k in names)) for k, v in params_np.items()}
    B = len(batch["e1"])
    H = int(md.get("emb_h", 10))
    d, r = int(md["ent_emb_size"]), int(md["rel_emb_size"])
    C = int(md.get("conv_num_channels", 32))
    in_h = H * 2 if (md.get("context_rel_out", None) is None and md.get("context_rel_conv", None) is None) else H
    fh, fw = int(md.get("conv_filter_height", 3)), int(md.get("conv_filter_width", 3))
    F = (in_h - fh + 1) * (d // H - fw + 1) * C
    kh = dropout_keep(seed, step, 1, B * F, float(md.get("hidden_dropout", 0.0)))
    ko = dropout_keep(seed, step, 2, B * d, float(md.get("output_dropout", 0.0)))
    kc = {}
    lk = md.get("do_parameter_lookup", False)
    ctx_o = [] if lk else (md.get("context_rel_out", None) or [])
    ctx_c = [] if lk else (md.get("context_rel_conv", None) or [])
    for gi, (g, ctx) in enumerate((("fc_weights", ctx_o), ("fc_bias", ctx_o), ("conv1_weights", ctx_c), ("conv1_bias", ctx_c))):
        for i, n in enumerate(ctx):
            kc[(g, i)] = dropout_keep(seed, step, 16 + 8 * gi + i, B * int(n), float(md.get("context_rel_dropout", 0.0)))
    loss, stats, _, _ = forward_train(p, md, batch, kh, ko, keep_ctx=kc)
    loss.backward()
    grads = {k: p[k].grad.numpy().copy() for k in names}
    for bn, (mean, var, n) in stats.items():
        unbiased = bn == "Conv1BN"                                # [TF-semantics] fused 4-D kernel vs 2-D fallback (FCBN, generators)
        var_m = var.numpy() * (n / (n - 1.0)) if unbiased else var.numpy()
        params_np[bn + "/moving_mean"] = params_np[bn + "/moving_mean"] * momentum + mean.numpy() * (1.0 - momentum)
        params_np[bn + "/moving_variance"] = params_np[bn + "/moving_variance"] * momentum + var_m * (1.0 - momentum)
    tr = {k: np.asarray(params_np[k], np.float64) for k in names}
    gn = opt.step(tr, grads)
    for k in names:
        params_np[k] = tr[k]
    return float(loss.detach()), grads, gn
