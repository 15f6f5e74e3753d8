"""CPU: pins the oracle (oracle/coper_oracle.py) against the committed golden vectors.

rank_*.npz were produced by the REFERENCE'S OWN ranking_and_hits (metrics.py) under a stub
tensorflow module; cpg_substeps.npz by the reference's PyTorch sister implementation
(fact_network.py); see oracle/gen_golden.py."""
import os

import numpy as np
import pytest

from coper_amd import data as cdata
from coper_amd.metrics import hits_and_means
from oracle import coper_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.mark.parametrize("E", [14, 257, 4099])
def test_ranker_matches_reference_ranking_and_hits(golden_dir, E):
    g = _load(golden_dir, "rank_E%d.npz" % E)
    pred, e2, ip, ix = g["pred"], g["e2"], g["filt_indptr"], g["filt_idx"]
    # closed form
    ng, ne = O.rank_counts(pred, e2, ip, ix)
    assert np.all(ne == 0)
    ranks = 1 + ng
    assert np.array_equal(ranks, g["closed_form_rank"])
    mr, mrr, hits = O.metrics_from_ranks(ranks, tuple(int(k) for k in g["ref_hits_k"]))
    assert mr == float(g["ref_mr"])                       # float64 mean of ints: exact
    assert abs(mrr - float(g["ref_mrr"])) <= 1e-16 * 4
    for k, v in zip(g["ref_hits_k"], g["ref_hits"]):
        assert hits[int(k)] == float(v)
    # literal restatement (dense mask, argsort)
    e2_multi = cdata.csr_to_dense_filter(ip, ix, E)
    assert np.array_equal(O.rank_dense_literal(pred, e2, e2_multi), ranks)
    # product-side host metric code (coper_amd.metrics.hits_and_means) gives the same numbers
    mr2, mrr2, hits2 = hits_and_means(ranks, tuple(int(k) for k in g["ref_hits_k"]))
    assert mr2 == float(g["ref_mr"]) and abs(mrr2 - float(g["ref_mrr"])) <= 4e-16
    assert all(hits2[int(k)] == float(v) for k, v in zip(g["ref_hits_k"], g["ref_hits"]))
    # C restatement
    ngc, nec = O.rank_counts_c(pred, e2, ip, ix)
    assert np.array_equal(ngc, ng) and np.array_equal(nec, ne)


def test_ranker_tie_band(golden_dir, oracle_chain):
    g = _load(golden_dir, "rank_ties.npz")
    ng, ne = O.rank_counts(g["pred"], g["e2"], g["filt_indptr"], g["filt_idx"])
    assert np.array_equal(ng, g["n_greater"]) and np.array_equal(ne, g["n_equal"])
    # the reference's rank under ties is argsort-order dependent: it must lie inside the band
    assert np.all(g["ref_rank"] >= 1 + ng) and np.all(g["ref_rank"] <= 1 + ng + ne)
    ngc, nec = O.rank_counts_c(g["pred"], g["e2"], g["filt_indptr"], g["filt_idx"])
    assert np.array_equal(ngc, ng) and np.array_equal(nec, ne)


@pytest.mark.parametrize("tag,hidden", [("lin", []), ("mlp", [16])])
def test_generator_substeps_match_minerva_torch(golden_dir, tag, hidden):
    g = _load(golden_dir, "cpg_substeps.npz")
    R, X, E2, b = g[tag + "_R"], g[tag + "_X"], g[tag + "_E2"], g[tag + "_b"]
    F, d = X.shape[1], E2.shape[1]
    md = dict(cdata._COMMON, num_ent=E2.shape[0], num_rel=4, ent_emb_size=d, rel_emb_size=R.shape[1], emb_h=10, emb_w=d // 10,
              context_rel_conv=None, context_rel_out=hidden, context_rel_use_batch_norm=False)
    dims = O.Dims(md)
    p = {}
    for i in range(int(g[tag + "_nproj"])):
        p["fc_weights/CPG/Projection%d" % i] = g["%s_Pw%d" % (tag, i)]
        p["fc_bias/CPG/Projection%d" % i] = g["%s_Pb%d" % (tag, i)]
    W = O.generate(R, p, "fc_weights", dims, hidden).reshape(-1, F, d)
    np.testing.assert_allclose(W, g[tag + "_W"], rtol=1e-5, atol=1e-6)
    fc = np.matmul(X[:, None, :], W)[:, 0, :] + O.generate(R, p, "fc_bias", dims, hidden)
    np.testing.assert_allclose(fc, g[tag + "_fc"], rtol=1e-5, atol=2e-5)
    S = O.score_all(np.maximum(fc, 0), E2, b)
    np.testing.assert_allclose(S, g[tag + "_S"], rtol=1e-5, atol=2e-5)


def test_conv_bn_relu_matches_torch_functional(golden_dir):
    g = _load(golden_dir, "conv_torch.npz")
    y = O.conv2d_valid_nhwc(g["img"], g["K"].reshape(3, 3, -1), False) + g["kb"]
    p = {"Conv1BN/gamma": g["bn_gamma"], "Conv1BN/beta": g["bn_beta"], "Conv1BN/moving_mean": g["bn_mean"],
         "Conv1BN/moving_variance": g["bn_var"]}
    y = np.maximum(O._bn(y, p, "Conv1BN", np.float32), 0)
    np.testing.assert_allclose(y, g["out_nhwc"], rtol=1e-5, atol=1e-5)


def test_kat_cross_correlation_no_flip():
    # hand-written: 3x3 cross-correlation on an arange image (tf.nn.conv2d does NOT flip the filter)
    img = np.arange(20, dtype=np.float32).reshape(1, 4, 5)
    K = np.zeros((3, 3, 2), np.float32)
    K[0, 0, 0] = 1.0          # picks img[i, j]
    K[2, 1, 1] = 2.0          # picks 2 * img[i+2, j+1]
    y = O.conv2d_valid_nhwc(img, K, False)
    assert y.shape == (1, 2, 3, 2)
    assert np.array_equal(y[0, :, :, 0], img[0, 0:2, 0:3])
    assert np.array_equal(y[0, :, :, 1], 2 * img[0, 2:4, 1:4])


def test_kat_bn_eps_and_flatten_order():
    p = {"B/gamma": np.array([2.0], np.float32), "B/beta": np.array([0.5], np.float32),
         "B/moving_mean": np.array([1.0], np.float32), "B/moving_variance": np.array([4.0 - 1e-3], np.float32)}
    out = O._bn(np.array([[3.0]], np.float32), p, "B", np.float32)
    np.testing.assert_allclose(out, [[2.0 * (3.0 - 1.0) / 2.0 + 0.5]], rtol=1e-6)
    # NHWC flatten: index (i*Wo + j)*C + c  (models.py:404)
    y = np.arange(2 * 3 * 4, dtype=np.float32).reshape(1, 2, 3, 4)
    flat = y.reshape(1, -1)
    assert flat[0, (1 * 3 + 2) * 4 + 3] == y[0, 1, 2, 3]


def test_kat_rank_closed_form():
    pred = np.array([[0.1, 0.9, 0.5, 0.7]], np.float32)
    e2 = np.array([2])
    # filter removes entity 1 (score 0.9); entity 3 (0.7) still beats the target 0.5 -> rank 2
    ng, ne = O.rank_counts(pred, e2, np.array([0, 2]), np.array([1, 2]))
    assert (1 + ng[0], ne[0]) == (2, 0)
    ng, ne = O.rank_counts(pred, e2, np.array([0, 0]), np.zeros(0, np.int64))
    assert 1 + ng[0] == 3


@pytest.mark.parametrize("name", ["plain", "cpg_fc", "cpg_fc_mlp", "cpg_conv_fc", "cpg_conv_only_concat", "lookup"])
def test_forward_fixtures_regenerate(golden_dir, name):
    """The committed stage outputs are what the oracle computes from the committed weights (guards the
    fixtures against drifting from the oracle), and materialised == per-relation generated dense."""
    from oracle.gen_golden import FWD_CASES
    g = _load(golden_dir, "fwd_%s.npz" % name)
    md = dict(cdata._COMMON)
    md.update(FWD_CASES[name][0])
    p = {k[6:]: g[k] for k in g.files if k.startswith("param:")}
    q = {k[2:]: g[k] for k in g.files if k.startswith("q:")}
    st = O.forward(p, md, q["e1"], q["rel"], np.float32, materialise=True)
    np.testing.assert_allclose(st["h"], g["f32:h"], rtol=0, atol=1e-6)
    st2 = O.forward(p, md, q["e1"], q["rel"], np.float32, materialise=False)
    np.testing.assert_allclose(st2["h"], st["h"], rtol=1e-5, atol=2e-5)
    lg = O.score_all(st["h"], p["ent_emb"], p["pred_bias"])
    np.testing.assert_allclose(lg, g["f64:logits"], rtol=0, atol=1e-4)
    assert set(p) == set(cdata.param_shapes(md))
    for k, shape in cdata.param_shapes(md).items():
        assert p[k].shape == tuple(shape), k


def test_chain_oracle_close_to_blas(oracle_chain):
    rng = np.random.default_rng(0)
    h = rng.standard_normal((7, 200)).astype(np.float32)
    E = (rng.standard_normal((301, 200)) * 0.3).astype(np.float32)
    b = (rng.standard_normal(301) * 0.1).astype(np.float32)
    s = O.score_chain(h, E, b)
    ref = h.astype(np.float64) @ E.astype(np.float64).T + b
    assert np.abs(s - ref).max() < 2e-5
    # d not a multiple of 8
    s2 = O.score_chain(h[:, :13], E[:, :13], b)
    ref2 = h[:, :13].astype(np.float64) @ E[:, :13].astype(np.float64).T + b
    assert np.abs(s2 - ref2).max() < 1e-5


@pytest.mark.parametrize("tag", ["plain", "cpg"])
def test_end_to_end_matches_minerva_torch_models(golden_dir, tag):
    """conv -> dense -> score of the oracle against the OUTPUT of the reference's PyTorch sister models
    (fact_network.py ConvE.forward / CPG_ConvE.forward and forward_fact) on the mapped weights."""
    from tests.minerva_map import load_case, sigmoid
    g = _load(golden_dir, "minerva_e2e.npz")
    md, p, q, S, S_fact = load_case(g, tag)
    assert set(p) == set(cdata.param_shapes(md))
    st = O.forward(p, md, q["e1"], q["rel"], np.float32)
    logits = O.score_all(st["h"], p["ent_emb"], p["pred_bias"])
    assert np.abs(sigmoid(logits) - S).max() < 2e-6
    fact = O.score_lookup(st["h"], p["ent_emb"], p["pred_bias"], q["e2"][:, None])
    assert np.abs(sigmoid(fact) - S_fact).max() < 2e-6
    st64 = O.forward(p, md, q["e1"], q["rel"], np.float64)
    assert np.abs(sigmoid(O.score_all(st64["h"], p["ent_emb"], p["pred_bias"])) - S).max() < 2e-6


@pytest.mark.parametrize("name", ["plain", "cpg_fc", "cpg_fc_mlp", "cpg_conv_fc", "cpg_conv_only_concat", "lookup"])
def test_torch_cpu_baseline_matches_the_oracle(golden_dir, name):
    """bench.py's cpu_baseline (oracle/coper_oracle_torch.py: torch-CPU conv2d / bmm / mm + the literal argsort
    ranker) computes what the NumPy oracle computes: stage output `h`, logits, and ranks of a filtered pass."""
    from oracle.coper_oracle_torch import TorchCPUModel
    from oracle.gen_golden import FWD_CASES
    g = _load(golden_dir, "fwd_%s.npz" % name)
    md = dict(cdata._COMMON)
    md.update(FWD_CASES[name][0])
    p = {k[6:]: g[k] for k in g.files if k.startswith("param:")}
    q = {k[2:]: g[k] for k in g.files if k.startswith("q:")}
    tm = TorchCPUModel(p, md)
    h = tm.forward(q["e1"], q["rel"])
    np.testing.assert_allclose(h.numpy(), g["f32:h"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(tm.predictions_all(h).numpy(), g["f64:logits"], rtol=0, atol=1e-4)
    # a filtered pass in two ragged batches against the oracle's closed form on the fp64 logits (the fixture's scores
    # are tie-free with margins far above the fp32 error)
    qq = cdata.synthetic_queries(md, len(q["e1"]), seed=3)
    ranks = tm.eval_pass(q["e1"], q["rel"], qq["e2"], qq["filt_indptr"], qq["filt_idx"], batch_size=max(1, len(q["e1"]) // 2 + 1))
    ng, ne = O.rank_counts(g["f64:logits"], qq["e2"], qq["filt_indptr"], qq["filt_idx"])
    assert np.all(ne == 0)       # tie-free (the target itself is not counted)
    assert np.array_equal(ranks, 1 + ng)
