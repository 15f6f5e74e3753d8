"""CPU oracle for the CoPER-ConvE hot path.  TEST INFRASTRUCTURE ONLY.

This module is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``coper_amd/`` imports it.

It restates, in NumPy, the arithmetic of the reference (paths relative to
``/root/reference/CoPER_ConvE/qa_cpg``):

* ``models.py:32-76``   ContextualParameterGenerator.generate  -> ``generate``
* ``models.py:79-94``   ParameterLookup.generate               -> ``lookup_params``
* ``models.py:176-180`` entity / relation gathers              -> ``forward``
* ``models.py:354-426`` ConvE._create_predictions              -> ``forward``
* ``models.py:428-446`` ConvE._compute_likelihoods             -> ``score_all`` / ``score_lookup``
* ``metrics.py:44-57``  filtered ranking                       -> ``rank_dense_literal`` / ``rank_counts``
* ``metrics.py:65-76``  Hits@k / MR / MRR means                -> ``metrics_from_ranks``

Pinning status (see DESIGN.md "Oracle"):
  * the ranker half is pinned against the reference's own ``metrics.py`` run in the
    authoring container under a stub ``tensorflow`` module
    (``oracle/gen_golden.py`` -> ``tests/golden/rank_*.npz``);
  * the generator / generated-dense / score sub-steps are pinned against the
    reference's PyTorch sister implementation ``CoPER_MINERVA/src/emb/fact_network.py``
    (``tests/golden/cpg_substeps.npz``);
  * conv -> dense -> score END TO END is pinned against the OUTPUT of the reference's PyTorch
    sister models run here on CPU (``fact_network.py`` ``ConvE.forward`` / ``CPG_ConvE.forward``
    and ``forward_fact``; ``tests/golden/minerva_e2e.npz``, weight mapping in
    ``tests/minerva_map.py``): sigmoid scores agree to 2e-6;
  * the TF-1.14 graph itself (``models.py``) cannot be executed here (TensorFlow is an
    un-vendored, loosely pinned dependency: ``requirements.txt:6``, "tensorflow-gpu==1.14"
    in ``CoPER_ConvE/README.md:115-116``) and the reference ships no tests, so what stays
    restated-from-semantics only is the TF-specific residue: BN placement after the conv
    (``Conv1BN``, the sister model has none), eps = 1e-3, and the NHWC flatten order --
    each covered by a hand-written KAT and a ``torch.nn.functional`` cross-check.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np

BN_EPS = 1e-3  # tf.layers.batch_normalization default epsilon (models.py:63,386,416 pass none)

BN_LEAVES = ("gamma", "beta", "moving_mean", "moving_variance")


# --------------------------------------------------------------------------------------
# configuration handling (models.py:98-130, 203-336)
# --------------------------------------------------------------------------------------
class Dims(object):
    """Shapes derived exactly as ``ConvE._create_variables`` derives them
    (models.py:261-271), plus the explicit (emb_h, emb_w) the build adds."""

    def __init__(self, md: dict):
        self.num_ent = int(md["num_ent"])
        self.num_rel = int(md["num_rel"])
        self.d = int(md["ent_emb_size"])
        self.r = int(md["rel_emb_size"])
        self.lookup = bool(md.get("do_parameter_lookup", False))
        self.fh = int(md.get("conv_filter_height", 3))
        self.fw = int(md.get("conv_filter_width", 3))
        self.C = int(md.get("conv_num_channels", 32))
        self.concat_rel = bool(md.get("concat_rel", False))
        self.ctx_conv = md.get("context_rel_conv", None)
        self.ctx_out = md.get("context_rel_out", None)
        self.ctx_bn = bool(md.get("context_rel_use_batch_norm", False))
        self.gen_conv = self.ctx_conv is not None
        self.gen_fc = self.ctx_out is not None
        # models.py:360 -- e1 is stacked on the reshaped relation only for plain ConvE
        self.stacked = (not self.gen_conv) and (not self.gen_fc) and (not self.lookup)
        # models.py:261-262,355 hard-code (10, d // 10); the build exposes it.
        self.emb_h = int(md.get("emb_h", 10))
        self.emb_w = int(md.get("emb_w", self.d // self.emb_h))
        if self.emb_h * self.emb_w != self.d:
            raise ValueError("emb_h * emb_w must equal ent_emb_size")
        self.in_h = self.emb_h
        if self.stacked:
            if self.r % self.emb_h != 0 or self.r // self.emb_h != self.emb_w:
                raise ValueError("plain ConvE stacks e1 on rel: needs rel_emb_size == ent_emb_size")
            self.in_h = 2 * self.emb_h
        elif (not self.gen_conv) and (not self.gen_fc):
            # models.py:263-264 adds the rows but :360 does not stack -> shape error in TF.
            raise ValueError("do_parameter_lookup with both contexts None is ill-formed in the reference")
        self.in_w = self.emb_w
        self.Ho = self.in_h - self.fh + 1
        self.Wo = self.in_w - self.fw + 1
        self.F_conv = self.Ho * self.Wo * self.C
        self.F = self.F_conv + (self.r if self.concat_rel else 0)
        if self.lookup and self.concat_rel:
            raise ValueError("g_lookup passes relation ids, cannot concat_rel (models.py:180,406)")


def _bn(x, p, prefix, dtype):
    g = p[prefix + "/gamma"].astype(dtype)
    b = p[prefix + "/beta"].astype(dtype)
    m = p[prefix + "/moving_mean"].astype(dtype)
    v = p[prefix + "/moving_variance"].astype(dtype)
    inv = g / np.sqrt(v + dtype(BN_EPS))
    return x * inv + (b - m * inv)


def generate(context, p, name, dims: Dims, hidden: Sequence[int], dtype=np.float32):
    """models.py:56-76 in inference mode: chain of bias-free projections; hidden layers get
    (optional) BN -> ReLU -> dropout(identity); the last projection is bare."""
    v = context.astype(dtype)
    n_proj = len(hidden) + 1
    for i in range(n_proj - 1):
        v = v @ p["%s/CPG/Projection%d" % (name, i)].astype(dtype)
        if dims.ctx_bn:
            v = _bn(v, p, "%s/CPG/Projection%d/BatchNorm" % (name, i), dtype)
        v = np.maximum(v, dtype(0))
    return v @ p["%s/CPG/Projection%d" % (name, n_proj - 1)].astype(dtype)


def lookup_params(rel_ids, p, name, dtype=np.float32):
    """models.py:90-94: row gather from the [num_rel, prod(shape)] table."""
    return p[name][np.asarray(rel_ids)].astype(dtype)


def per_relation_params(p, md, dtype=np.float32):
    """Generated / looked-up parameters for EVERY relation id (what the build caches in
    ``coper_prepare``): returns dict with conv_w [R,fh,fw,C] or None, conv_b [R,C] or None,
    fc_w [R,F,d] or None, fc_b [R,d] or None."""
    dims = Dims(md)
    R = dims.num_rel
    out = {"conv_w": None, "conv_b": None, "fc_w": None, "fc_b": None}
    ids = np.arange(R)
    if dims.gen_conv:
        if dims.lookup:
            out["conv_w"] = lookup_params(ids, p, "conv1_weights", dtype).reshape(R, dims.fh, dims.fw, dims.C)
            out["conv_b"] = lookup_params(ids, p, "conv1_bias", dtype).reshape(R, dims.C)
        else:
            c = p["rel_emb"]
            out["conv_w"] = generate(c, p, "conv1_weights", dims, dims.ctx_conv, dtype).reshape(R, dims.fh, dims.fw, dims.C)
            out["conv_b"] = generate(c, p, "conv1_bias", dims, dims.ctx_conv, dtype).reshape(R, dims.C)
    if dims.gen_fc:
        if dims.lookup:
            out["fc_w"] = lookup_params(ids, p, "fc_weights", dtype).reshape(R, dims.F, dims.d)
            out["fc_b"] = lookup_params(ids, p, "fc_bias", dtype).reshape(R, dims.d)
        else:
            c = p["rel_emb"]
            out["fc_w"] = generate(c, p, "fc_weights", dims, dims.ctx_out, dtype).reshape(R, dims.F, dims.d)
            out["fc_b"] = generate(c, p, "fc_bias", dims, dims.ctx_out, dtype).reshape(R, dims.d)
    return out


def conv2d_valid_nhwc(img, K, per_sample):
    """Cross-correlation, no flip, VALID, stride 1 (tf.nn.conv2d; models.py:375-385).
    img [B,H,W]; K [fh,fw,C] or [B,fh,fw,C] -> [B,Ho,Wo,C]."""
    B, H, W = img.shape
    fh, fw = (K.shape[1], K.shape[2]) if per_sample else (K.shape[0], K.shape[1])
    Ho, Wo = H - fh + 1, W - fw + 1
    C = K.shape[-1]
    y = np.zeros((B, Ho, Wo, C), dtype=img.dtype)
    for u in range(fh):
        for v in range(fw):
            patch = img[:, u:u + Ho, v:v + Wo]
            if per_sample:
                y += patch[..., None] * K[:, u, v, :][:, None, None, :]
            else:
                y += patch[..., None] * K[u, v, :]
    return y


def forward(p: Dict[str, np.ndarray], md: dict, e1, rel, dtype=np.float32, materialise=True):
    """Inference-mode forward of ``ConvE`` up to ``predicted_e2_emb`` (models.py:176-183,
    354-426; recipe SURVEY 8-A steps 1-7).  Returns a dict of every stage.

    ``materialise=True`` computes the generated dense layer the reference way: build the
    per-sample weight tensor [B,F,d] (models.py:70,350) and contract it with a batched
    matmul (models.py:412).  ``False`` uses the algebraically identical per-relation form
    (used only to keep big test cases inside memory)."""
    dims = Dims(md)
    dt = dtype
    e1 = np.asarray(e1, dtype=np.int64)
    rel = np.asarray(rel, dtype=np.int64)
    B = e1.shape[0]
    st = {}
    x0 = p["ent_emb"][e1].astype(dt)                                     # models.py:176
    c = None if dims.lookup else p["rel_emb"][rel].astype(dt)            # models.py:177-180
    img = x0.reshape(B, dims.emb_h, dims.emb_w)                          # models.py:355
    if dims.stacked:                                                     # models.py:360-362
        img = np.concatenate([img, c.reshape(B, dims.emb_h, dims.r // dims.emb_h)], axis=1)
    st["img"] = img

    # --- conv stage (models.py:372-391) ---
    if dims.gen_conv:
        if dims.lookup:
            K = lookup_params(rel, p, "conv1_weights", dt).reshape(B, dims.fh, dims.fw, dims.C)
            kb = lookup_params(rel, p, "conv1_bias", dt).reshape(B, dims.C)
        else:
            K = generate(c, p, "conv1_weights", dims, dims.ctx_conv, dt).reshape(B, dims.fh, dims.fw, dims.C)
            kb = generate(c, p, "conv1_bias", dims, dims.ctx_conv, dt).reshape(B, dims.C)
        y = conv2d_valid_nhwc(img, K, True) + kb[:, None, None, :]
    else:
        K = p["conv1_weights"].astype(dt).reshape(dims.fh, dims.fw, dims.C)   # HWIO with I == 1
        y = conv2d_valid_nhwc(img, K, False) + p["conv1_bias"].astype(dt)
    st["conv"] = y
    y = _bn(y, p, "Conv1BN", dt)
    y = np.maximum(y, dt(0))
    st["conv_act"] = y

    # --- dense stage (models.py:400-419) ---
    x = y.reshape(B, -1)                                                 # (i, j, o) order, models.py:404
    if dims.concat_rel:
        x = np.concatenate([x, c], axis=1)                               # models.py:406-407
    st["x"] = x
    if dims.gen_fc:
        if dims.lookup:
            get_w = lambda ids: lookup_params(ids, p, "fc_weights", dt).reshape(-1, dims.F, dims.d)
            bvec = lookup_params(rel, p, "fc_bias", dt).reshape(B, dims.d)
            ctx = rel
        else:
            get_w = lambda cc: generate(cc, p, "fc_weights", dims, dims.ctx_out, dt).reshape(-1, dims.F, dims.d)
            bvec = generate(c, p, "fc_bias", dims, dims.ctx_out, dt).reshape(B, dims.d)
            ctx = c
        if materialise:
            W = get_w(ctx)                                               # [B,F,d]  models.py:350
            z = np.matmul(x[:, None, :], W)[:, 0, :] + bvec              # models.py:412
        else:
            z = np.empty((B, dims.d), dtype=dt)
            for rid in np.unique(rel):
                sel = np.nonzero(rel == rid)[0]
                Wr = get_w(ctx[sel[:1]])[0]
                z[sel] = x[sel] @ Wr
            z = z + bvec
    else:
        z = x @ p["fc_weights"].astype(dt) + p["fc_bias"].astype(dt)     # models.py:410
    st["z"] = z
    z = _bn(z, p, "FCBN", dt)                                            # models.py:416-418
    h = np.maximum(z, dt(0))                                             # models.py:419
    st["h"] = h
    return st


def score_all(h, ent_emb, pred_bias):
    """models.py:434-437: logits, no sigmoid."""
    return h @ ent_emb.astype(h.dtype).T + pred_bias.astype(h.dtype)


def score_lookup(h, ent_emb, pred_bias, lookup):
    """models.py:439-443: sampled scorer."""
    lookup = np.asarray(lookup)
    g = ent_emb[lookup].astype(h.dtype)                                  # [B,L,d]
    return np.einsum("bd,bld->bl", h, g) + pred_bias[lookup].astype(h.dtype)


# --------------------------------------------------------------------------------------
# ranker (metrics.py)
# --------------------------------------------------------------------------------------
def rank_dense_literal(pred, e2, e2_multi):
    """metrics.py:44-50 literally: mask known answers to -inf, restore the target,
    full argsort per row.  Mutates a copy.  Tie order = whatever np.argsort returns."""
    pred = np.array(pred, copy=True)
    e2 = np.asarray(e2)
    rows = np.arange(0, len(pred))
    target_values = pred[rows, e2]
    pred[e2_multi == 1] = -np.inf
    pred[rows, e2] = target_values
    ranks = np.empty(len(pred), dtype=np.int64)
    for i in range(len(pred)):
        args = np.argsort(-pred[i])
        ranks[i] = int(np.where(args == e2[i])[0][0]) + 1
    return ranks


def dense_to_csr(e2_multi):
    """Dense 0/1 filter mask (data.py:182-186) -> CSR (indptr int64 [B+1], idx int64)."""
    B = e2_multi.shape[0]
    indptr = np.zeros(B + 1, dtype=np.int64)
    idx = []
    for i in range(B):
        nz = np.nonzero(e2_multi[i] == 1)[0]
        idx.append(nz)
        indptr[i + 1] = indptr[i] + len(nz)
    return indptr, (np.concatenate(idx).astype(np.int64) if idx else np.zeros(0, np.int64))


def rank_counts(pred, e2, indptr, idx):
    """Closed form of metrics.py:44-50: over unfiltered j != e2, count scores strictly
    greater than / equal to the target.  rank is any value in
    [1 + n_greater, 1 + n_greater + n_equal] (unstable argsort); == 1 + n_greater when
    tie-free."""
    B, _ = pred.shape
    e2 = np.asarray(e2)
    ng = np.empty(B, dtype=np.int64)
    ne = np.empty(B, dtype=np.int64)
    for i in range(B):
        t = pred[i, e2[i]]
        keep = np.ones(pred.shape[1], dtype=bool)
        keep[idx[indptr[i]:indptr[i + 1]]] = False
        keep[e2[i]] = False
        row = pred[i][keep]
        ng[i] = int(np.count_nonzero(row > t))
        ne[i] = int(np.count_nonzero(row == t))
    return ng, ne


def metrics_from_ranks(ranks, hits_to_compute=(1, 3, 5, 10, 20)):
    """metrics.py:53-57,65-76: float64 means of per-query 1.0/0.0 hits, ranks, 1/ranks."""
    ranks = list(int(r) for r in ranks)
    hits = {}
    for k in hits_to_compute:
        hits[k] = np.mean([1.0 if r <= k else 0.0 for r in ranks])
    mr = np.mean(ranks)
    mrr = np.mean(1. / np.array(ranks))
    return mr, mrr, hits


def topk_filtered(pred, e2, indptr, idx, k):
    """Top-k of the filtered row (known answers except the target masked out), order
    (score desc, id asc).  Not in the reference ranker; used by the multi-GPU exchange."""
    B, N = pred.shape
    vals = np.full((B, k), -np.inf, dtype=pred.dtype)
    ids = np.full((B, k), -1, dtype=np.int64)
    for i in range(B):
        row = np.array(pred[i], copy=True)
        t = row[e2[i]]
        row[idx[indptr[i]:indptr[i + 1]]] = -np.inf
        row[e2[i]] = t
        order = np.lexsort((np.arange(N), -row))[:k]
        order = order[np.isfinite(row[order]) | (row[order] > -np.inf)]
        vals[i, :len(order)] = row[order]
        ids[i, :len(order)] = order
    return vals, ids


# --------------------------------------------------------------------------------------
# reference-semantics CPU baseline (bench.py cpu_baseline leg, kind == "port")
# --------------------------------------------------------------------------------------
def eval_pass_reference_semantics(p, md, e1, rel, e2, indptr, idx, batch_size=512):
    """One evaluation pass the way the reference runs it (run_cpg.py:18-35 ->
    metrics.py:38-60): per batch, forward with the generated dense weights materialised
    [B,F,d], logits for all entities, dense filter mask, per-row argsort."""
    dims = Dims(md)
    ranks = []
    Q = len(e1)
    for s in range(0, Q, batch_size):
        sl = slice(s, min(Q, s + batch_size))
        st = forward(p, md, e1[sl], rel[sl], np.float32, materialise=True)
        pred = score_all(st["h"], p["ent_emb"], p["pred_bias"])
        nb = pred.shape[0]
        e2_multi = np.zeros((nb, dims.num_ent), dtype=np.float32)        # data.py:182-186
        for i in range(nb):
            e2_multi[i, idx[indptr[s + i]:indptr[s + i + 1]]] = 1.0
        ranks.append(rank_dense_literal(pred, e2[sl], e2_multi))
    return np.concatenate(ranks)


# --------------------------------------------------------------------------------------
# C restatement (oracle/coper_oracle_chain.c): bit-exact score chain + fast rank counts
# --------------------------------------------------------------------------------------
_CHAIN = None


def chain_lib():
    """Loads oracle/_build/libcoper_oracle.so (built by `make -C oracle` / __graft_entry__.build())."""
    global _CHAIN
    if _CHAIN is None:
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libcoper_oracle.so")
        lib = C.CDLL(path)
        P = C.c_void_p
        lib.oracle_score_chain.argtypes = [P, P, P, C.c_int64, C.c_int64, C.c_int, P]
        lib.oracle_score_chain.restype = None
        lib.oracle_rank_counts.argtypes = [P, P, P, P, C.c_int64, C.c_int64, P, P]
        lib.oracle_rank_counts.restype = None
        _CHAIN = lib
    return _CHAIN


def score_chain(h, ent_emb, pred_bias):
    """Logits [B, N] by the exact fma chain of the HIP kernels (k-pairs (k, k+4), start = pred_bias)."""
    lib = chain_lib()
    h = np.ascontiguousarray(h, np.float32)
    E = np.ascontiguousarray(ent_emb, np.float32)
    b = np.ascontiguousarray(pred_bias, np.float32)
    out = np.empty((h.shape[0], E.shape[0]), np.float32)
    lib.oracle_score_chain(h.ctypes.data, E.ctypes.data, b.ctypes.data, h.shape[0], E.shape[0], h.shape[1], out.ctypes.data)
    return out


def rank_counts_c(pred, e2, indptr, idx):
    lib = chain_lib()
    pred = np.ascontiguousarray(pred, np.float32)
    e2 = np.ascontiguousarray(e2, np.int64)
    indptr = np.ascontiguousarray(indptr, np.int64)
    idx = np.ascontiguousarray(idx, np.int64)
    B, N = pred.shape
    ng = np.empty(B, np.int64)
    ne = np.empty(B, np.int64)
    lib.oracle_rank_counts(pred.ctypes.data, e2.ctypes.data, indptr.ctypes.data, idx.ctypes.data, B, N, ng.ctypes.data, ne.ctypes.data)
    return ng, ne
