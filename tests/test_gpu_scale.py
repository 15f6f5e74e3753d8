"""GPU: the bf16x3 ("x3", fp16-split) arithmetic is SCALE-INVARIANT (coper_amd/csrc/split16.h).

fp16's hi + lo carries 22 bits only inside a window of magnitudes; the reference initialises its tables far below it
(models.py:205-208: xavier_initializer, +-0.0202 at FB15k-237's shape, +-7.7e-4 for a 10M-entity table).  Every operand class is
therefore moved into the window by an exact power of two.  These tests pin what that buys, at every scale and through the C ABI:

  * ranks and tie counts of the x3 mode == the f32 mode's (the fp32 chain) on the same h, for EVERY query;
  * max |logit_x3 - logit_chain| over millions of logits <= a QUARTER of what the exact band allows one logit,
    kappa (|h_q| max|E_e| + 8 max|pred_bias|) with kappa = 1e-6 (bf16x3_chain.h: x3_band_tau) -- i.e. 2.5e-7 |h_q| max|E_e| where the
    products dominate (1.8e-7 on round 3's well-scaled N(0, 0.3^2) table; the fp32 chain itself is 2.3e-7 from float64);
  * the run-time band audit (coper_band_audit) stays below 0.5 of the band's allowance;
  * logits come back in the caller's units (within 1e-3 of float64, relative to the logits' own scale).
"""
import numpy as np
import pytest
import torch

from coper_amd import data as cdata

pytestmark = pytest.mark.gpu

KAPPA = 1e-6             # COPER_BAND_KAPPA_DEFAULT
BIAS_WEIGHT = 8.0        # X3_BAND_BIAS_WEIGHT
REL_ERR_BAR = 0.25       # max |s_x3 - s_chain| / (kappa (|h_q| max|E_e| + 8 max|bias|)): a quarter of the band's allowance
AUDIT_BAR = 0.5
H_REL_BAR = 1e-5         # max |h - h_float64| / max |h_float64| of the x3 encoder (measured 1.3e-5 ABSOLUTE on O(1) embeddings in round 3)


def _model(md, params, **kw):
    from coper_amd.models import ConvE
    if kw.get("score_mode") == "bf16x3":
        kw.setdefault("band_audit_period", 1)      # every count launch audited (the default spares 7 of 8 short launches)
    m = ConvE(md, device="cuda:0", **kw)
    m.load_parameters(params)
    m.prepare()
    return m


def _tables(kind, md, seed):
    """(params, description): the scoring operands of one scale scenario; the encoder weights stay the synthetic ones unless
    the scenario is the reference's own initialisation."""
    rng = np.random.default_rng(seed + 17)
    E, d = int(md["num_ent"]), int(md["ent_emb_size"])
    if kind == "xavier":                       # models.py:205-214,284-293: everything as _create_variables draws it
        p = cdata.reference_init_params(md, seed)
        return p
    p = cdata.synthetic_params(md, seed)
    if kind == "xavier_bias":                  # the reference's tables with a trained-looking pred_bias beside them
        r = cdata.reference_init_params(md, seed)
        p["ent_emb"] = r["ent_emb"]
        p["pred_bias"] = (rng.standard_normal(E) * 1e-3).astype(np.float32)
    elif kind == "n1e-2":
        p["ent_emb"] = (rng.standard_normal((E, d)) * 1e-2).astype(np.float32)
    elif kind == "n1e-3":
        p["ent_emb"] = (rng.standard_normal((E, d)) * 1e-3).astype(np.float32)
        p["pred_bias"] = (rng.standard_normal(E) * 1e-4).astype(np.float32)
    elif kind == "rowspread":                  # row norms spread over 10^3 (log-uniform)
        s = np.exp(rng.uniform(np.log(1e-3), 0.0, E)).astype(np.float32)
        p["ent_emb"] = (rng.standard_normal((E, d)).astype(np.float32) * 0.3 * s[:, None]).astype(np.float32)
    elif kind == "clamp":                      # elements at and beyond fp16's largest finite value (65,504)
        p["ent_emb"] = (rng.standard_normal((E, d)) * 3e4).astype(np.float32)
        p["ent_emb"][::97, 3] = 6.5e4
        p["ent_emb"][1::97, 5] = -1.3e5
        p["pred_bias"] = (rng.standard_normal(E) * 1e4).astype(np.float32)
    elif kind == "tiny":                       # far below fp16's subnormals without the scaling
        p["ent_emb"] = (rng.standard_normal((E, d)) * 1e-9).astype(np.float32)
        p["pred_bias"] = np.zeros(E, np.float32)
    elif kind != "n0.1":
        raise KeyError(kind)
    return p


def _pin_to_c_chain(O, p, h, q, l32_rows, r32, ne32, n_sample=48):
    """The checker behind the checker: at THIS operand scale, on a sample of the queries, the f32 mode's
    logits are the C restatement of the fp32 chain bit for bit, and its ranks / tie counts are the C closed form of
    metrics.py:44-50 on those logits -- no GPU arithmetic on the checking side."""
    Q = h.shape[0]
    sel = np.unique(np.linspace(0, Q - 1, min(Q, n_sample)).astype(np.int64))
    hs = np.ascontiguousarray(h[torch.as_tensor(sel, device=h.device)].cpu().numpy())
    chain = O.score_chain(hs, p["ent_emb"], p["pred_bias"])
    assert np.array_equal(l32_rows(sel), chain), "f32-mode logits differ from the C chain at this scale"
    ip, ix = q["filt_indptr"], q["filt_idx"]
    sip = np.zeros(len(sel) + 1, np.int64)
    sip[1:] = np.cumsum(ip[sel + 1] - ip[sel])
    six = np.concatenate([ix[ip[i]:ip[i + 1]] for i in sel]) if len(sel) else np.zeros(0, np.int64)
    ng_c, ne_c = O.rank_counts_c(chain, q["e2"][sel], sip, six)
    assert np.array_equal(r32.cpu().numpy()[sel], 1 + ng_c), "f32-mode ranks differ from the C oracle's at this scale"
    assert np.array_equal(ne32.cpu().numpy()[sel], ne_c)


def _check(md, p, Q, h_scale=1.0, seed=0, kappa=0.0, O=None):
    m3 = _model(md, p, score_mode="bf16x3", rank_band_kappa=kappa)
    m32 = _model(md, p, score_mode="f32")
    q = cdata.synthetic_queries(md, Q, seed=seed)
    h = m3.encode(q["e1"], q["rel"])
    # the x3 ENCODER at this scale: its operands (dense weights per relation, conv activations) are moved into fp16's window
    # too, so h is as close to float64 as at any other scale -- relative to the embedding's own magnitude
    from oracle.coper_oracle_torch import TorchCPUModel
    tm = TorchCPUModel(p, md, device=h.device, dtype=torch.float64)
    h64 = torch.cat([tm.forward(q["e1"][s:s + 256], q["rel"][s:s + 256]) for s in range(0, min(Q, 512), 256)])
    h_rel = float((h[:h64.shape[0]].double() - h64).abs().max() / h64.abs().max().clamp_min(1e-300))
    assert h_rel <= H_REL_BAR, h_rel
    _check.last_h_rel = h_rel
    if h_scale != 1.0:
        h = (h * h_scale).contiguous()
    m3.band_audit()                                                   # reset
    r3, ne3 = m3.rank(h, q["e2"], q["filt_indptr"], q["filt_idx"])
    r32, ne32 = m32.rank(h, q["e2"], q["filt_indptr"], q["filt_idx"])
    ratio, n_pairs = m3.band_audit()
    assert np.array_equal(r3.cpu().numpy(), r32.cpu().numpy()), "x3 ranks differ from the fp32 chain's"
    assert np.array_equal(ne3.cpu().numpy(), ne32.cpu().numpy())
    assert n_pairs > 0 and ratio <= AUDIT_BAR, (ratio, n_pairs)
    if O is not None:     # ... and the f32 mode itself is pinned to the C oracle at this scale
        _pin_to_c_chain(O, p, h, q, lambda sel: m32.score_all(h[torch.as_tensor(sel, device=h.device)].contiguous()).cpu().numpy(), r32, ne32)
    # the fused pass (coper_encode_rank) agrees when h is the model's own
    if h_scale == 1.0:
        rf, _ = m3.rank_pass(q["e1"], q["rel"], q["e2"], q["filt_indptr"], q["filt_idx"], want_equal=False)
        assert np.array_equal(rf.cpu().numpy(), r32.cpu().numpy())
        ratio_f, n_f = m3.band_audit()
        assert n_f > 0 and ratio_f <= AUDIT_BAR, (ratio_f, n_f)
    # the logits themselves: x3 against the chain (the f32 mode's logits ARE the chain, bit for bit: test_gpu_parity), relative
    # to the band's allowance
    sub = slice(0, min(Q, 384))
    l3 = m3.score_all(h[sub]).double()
    l32 = m32.score_all(h[sub]).double()
    hn = h[sub].double().norm(dim=1)
    E = torch.as_tensor(p["ent_emb"], device=h.device).double()
    emax = float(E.norm(dim=1).max())
    bmax = float(np.abs(p["pred_bias"]).max())
    allow = (KAPPA * (hn * emax + BIAS_WEIGHT * bmax)).clamp_min(1e-300)
    rel = float(((l3 - l32).abs().max(dim=1).values / allow).max())
    assert rel <= REL_ERR_BAR, rel
    # ... and against float64 in the caller's units
    l64 = torch.addmm(torch.as_tensor(p["pred_bias"], device=h.device).double(), h[sub].double(), E.t())
    scale = float(l64.abs().max())
    assert float((l3 - l64).abs().max()) <= 1e-3 * max(scale, 1e-30), (float((l3 - l64).abs().max()), scale)
    # target scores and the sampled scorer leave in the caller's units too, with the bits of score_all
    tg = m3.target_scores(h[sub], q["e2"][sub])
    assert np.array_equal(tg[0].cpu().numpy(), l3.float()[torch.arange(l3.shape[0]), torch.as_tensor(q["e2"][sub], device=h.device)].cpu().numpy())
    m3.close()
    m32.close()
    return rel, ratio, n_pairs


@pytest.mark.parametrize("kind", ["xavier", "xavier_bias", "n0.1", "n1e-2", "n1e-3", "rowspread", "clamp", "tiny"])
def test_x3_ranks_equal_chain_at_every_table_scale_fb15k237(kind, oracle_chain):
    """FB15k-237's shapes (|E| = 14,541, d = 200, 474 relations): every query's rank and tie count."""
    md = cdata.model_descriptors("fb15k237_cpg")
    p = _tables(kind, md, 0)
    rel, ratio, n = _check(md, p, 4096, O=oracle_chain)
    print("fb15k237 %-12s max |s_x3 - s_chain| = %.3f of the band's allowance over %d logits; band audit %.3f over %d pairs; "
          "encoder max |h - h64| / max |h64| = %.2e" % (kind, rel, 384 * md["num_ent"], ratio, n, _check.last_h_rel))


@pytest.mark.parametrize("h_scale", [1e-6, 1e-3, 1e3, 1e6])
def test_x3_ranks_equal_chain_at_every_query_scale(h_scale, oracle_chain):
    """The query side: the same embeddings multiplied by a constant (a model whose FCBN gamma is that much larger / smaller)."""
    md = cdata.model_descriptors("fb15k237_cpg", num_ent=6000, num_rel=40)
    p = cdata.synthetic_params(md, 3)
    _check(md, p, 1024, h_scale=h_scale, seed=5, O=oracle_chain)


def test_x3_mixed_query_magnitudes_in_one_batch():
    """One exponent serves a packed batch (its largest query); queries 2^20 smaller share it.  Their ranks stay the chain's:
    the band's absolute term (bf16x3_chain.h: x3_band_tau) widens their bands instead."""
    md = cdata.model_descriptors("fb15k237_cpg", num_ent=6000, num_rel=40)
    p = cdata.synthetic_params(md, 4)
    m3 = _model(md, p, score_mode="bf16x3")
    m32 = _model(md, p, score_mode="f32")
    q = cdata.synthetic_queries(md, 512, seed=6)
    h = m3.encode(q["e1"], q["rel"])
    s = torch.ones(512, device=h.device)
    s[::3] = 2.0 ** -20
    s[1::7] = 2.0 ** -30
    s[5] = 0.0                                       # a dead query among them
    h = (h * s[:, None]).contiguous()
    r3, ne3 = m3.rank(h, q["e2"], q["filt_indptr"], q["filt_idx"])
    r32, ne32 = m32.rank(h, q["e2"], q["filt_indptr"], q["filt_idx"])
    assert np.array_equal(r3.cpu().numpy(), r32.cpu().numpy()) and np.array_equal(ne3.cpu().numpy(), ne32.cpu().numpy())
    m3.close()
    m32.close()


@pytest.mark.parametrize("kind", ["xavier10m", "n1e-3"])
def test_x3_ranks_equal_chain_10m_shape_tables(kind, oracle_chain):
    """The 10M-entity config's shapes (d = 256, 16 x 16 image, r = 32) on a 300,000-row slice of the table, with the element
    scale the reference's initialiser gives the FULL 10M-row table (limit sqrt(6 / (10^7 + 256)) = 7.7e-4, models.py:205-208)."""
    md = cdata.model_descriptors("synth10m_cpg", num_ent=300000, num_rel=64)
    p = cdata.synthetic_params(md, 0)
    rng = np.random.default_rng(5)
    E, d = 300000, int(md["ent_emb_size"])
    if kind == "xavier10m":
        lim = np.sqrt(6.0 / (1e7 + d))
        p["ent_emb"] = rng.uniform(-lim, lim, (E, d)).astype(np.float32)
        p["pred_bias"] = np.zeros(E, np.float32)
    else:
        p["ent_emb"] = (rng.standard_normal((E, d)) * 1e-3).astype(np.float32)
    rel, ratio, n = _check(md, p, 1024, O=oracle_chain)
    print("10M-shape %-10s max |s_x3 - s_chain| = %.3f of the band's allowance; band audit %.3f over %d pairs" % (kind, rel, ratio, n))


def test_x3_shards_agree_through_the_table_wide_exponent():
    """Two shard handles of a table whose halves differ in magnitude by 2^12: with the table-wide maximum (load_parameters
    sets it when it sees the whole table) the mode's logits are the unsharded handle's bits; a hint below a shard's own maximum
    is refused."""
    from coper_amd import _lib
    md = cdata.model_descriptors("fb15k237_cpg", num_ent=3000, num_rel=16)
    p = cdata.synthetic_params(md, 9)
    p["ent_emb"][:1500] *= np.float32(2.0 ** -12)
    q = cdata.synthetic_queries(md, 200, seed=2)
    full = _model(md, p, score_mode="bf16x3")
    h = full.encode(q["e1"], q["rel"])
    lg = full.score_all(h).cpu().numpy()
    shards = [_model(md, p, score_mode="bf16x3", shard=(0, 1500)), _model(md, p, score_mode="bf16x3", shard=(1500, 3000))]
    got = torch.cat([s.score_all(h) for s in shards], dim=1).cpu().numpy()
    assert np.array_equal(got, lg)
    tgt = sum(s.target_scores(h, q["e2"]) for s in shards)
    ng = sum(s.rank_counts(h, tgt, q["e2"], q["filt_indptr"], q["filt_idx"])[0] for s in shards)
    ranks, _ = full.rank(h, q["e2"], q["filt_indptr"], q["filt_idx"])
    assert np.array_equal((1 + ng).cpu().numpy(), ranks.cpu().numpy())
    shards[1].set_x3_ent_absmax(1e-6)
    with pytest.raises(_lib.CoperError, match="x3_ent_absmax"):
        shards[1].prepare()
    for m in shards + [full]:
        m.close()
