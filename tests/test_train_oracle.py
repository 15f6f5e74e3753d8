"""CPU: the training oracle (oracle/coper_train_oracle.py) against the inference oracle, finite differences
and a hand-computed AMSGrad trace (utils/amsgrad.py:130-159 arithmetic)."""
import numpy as np
import pytest
import torch

from coper_amd import data as cdata
from oracle import coper_oracle as O
from oracle import coper_train_oracle as T

_MD = dict(num_ent=61, num_rel=4, ent_emb_size=40, rel_emb_size=8, emb_h=10, emb_w=4, conv_num_channels=8,
           context_rel_conv=None, context_rel_out=[])


def _setup(train_stats, plain=False):
    md = dict(cdata._COMMON)
    md.update(_MD)
    if plain:
        md.update(rel_emb_size=40, context_rel_out=None)
    md.update(batch_norm_train_stats=train_stats, hidden_dropout=0.25, output_dropout=0.1, label_smoothing_epsilon=0.1)
    p = cdata.synthetic_params(md, seed=2)
    rng = np.random.default_rng(1)
    B, L = 10, 7
    batch = dict(e1=rng.integers(0, 61, B), rel=rng.integers(0, 4, B), lookup=rng.integers(0, 61, (B, L)),
                 labels=(rng.random((B, L)) < 0.3).astype(np.float32))
    return md, p, batch


def test_eval_mode_forward_equals_inference_oracle():
    for plain in (False, True):
        md, p, batch = _setup(False, plain)
        md["hidden_dropout"] = md["output_dropout"] = 0.0
        pt = {k: torch.tensor(np.asarray(v, np.float64)) for k, v in p.items()}
        F = O.Dims(md).F
        _, _, h, s = T.forward_train(pt, md, batch, np.ones(10 * F, np.float32), np.ones(10 * 40, np.float32))
        ref = O.forward(p, md, batch["e1"], batch["rel"], np.float64)
        assert np.abs(h.numpy() - ref["h"]).max() < 1e-12
        want = O.score_lookup(ref["h"], np.asarray(p["ent_emb"], np.float64), np.asarray(p["pred_bias"], np.float64), batch["lookup"])
        assert np.abs(s.numpy() - want).max() < 1e-12


def test_gradients_match_finite_differences():
    md, p, batch = _setup(True)
    B, F, d = 10, O.Dims(md).F, 40
    kh = T.dropout_keep(3, 0, 1, B * F, md["hidden_dropout"])
    ko = T.dropout_keep(3, 0, 2, B * d, md["output_dropout"])
    names = T.trainable_names(md)

    def loss_of(pp):
        pt = {k: torch.tensor(np.asarray(v, np.float64)) for k, v in pp.items()}
        return float(T.forward_train(pt, md, batch, kh, ko)[0])

    pt = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=(k in names)) for k, v in p.items()}
    loss, _, _, _ = T.forward_train(pt, md, batch, kh, ko)
    loss.backward()
    rng = np.random.default_rng(0)
    for leaf in ("rel_emb", "conv1_weights", "fc_weights/CPG/Projection0", "FCBN/gamma", "ent_emb"):
        g = pt[leaf].grad.numpy()
        flat = np.flatnonzero(np.abs(g.ravel()) > 1e-8)
        for idx in rng.choice(flat, size=min(4, len(flat)), replace=False):
            q = {k: np.array(v, np.float64) for k, v in p.items()}
            eps = 1e-6
            q[leaf].ravel()[idx] += eps
            up = loss_of(q)
            q[leaf].ravel()[idx] -= 2 * eps
            dn = loss_of(q)
            fd = (up - dn) / (2 * eps)
            assert abs(fd - g.ravel()[idx]) < 1e-6 + 1e-4 * abs(fd), (leaf, idx, fd, g.ravel()[idx])


def test_amsgrad_trace_by_hand():
    p = {"w": np.array([1.0, -2.0])}
    opt = T.AMSGrad(["w"], p, lr=0.1, beta1=0.9, beta2=0.999, eps=1e-8, clip=5.0)
    g1 = {"w": np.array([3.0, 4.0])}                                # norm 5: not clipped
    opt.step(p, g1)
    lr_t = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9)
    m, v = 0.1 * g1["w"], 0.001 * g1["w"] ** 2
    want = np.array([1.0, -2.0]) - lr_t * m / (np.sqrt(v) + 1e-8)
    assert np.allclose(p["w"], want, rtol=0, atol=1e-15)
    g2 = {"w": np.array([30.0, 40.0])}                              # norm 50: scaled by 0.1 -> (3, 4)
    opt.step(p, g2)
    lr_t2 = 0.1 * np.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    m2, v2 = 0.9 * m + 0.1 * g1["w"], 0.999 * v + 0.001 * g1["w"] ** 2
    vh2 = np.maximum(v, v2)
    assert np.allclose(p["w"], want - lr_t2 * m2 / (np.sqrt(vh2) + 1e-8), rtol=0, atol=1e-15)


def test_dropout_keep_rate_and_determinism():
    k = T.dropout_keep(9, 4, 1, 1 << 18, 0.3)
    assert abs(k.mean() - 0.7) < 5e-3
    assert np.array_equal(k, T.dropout_keep(9, 4, 1, 1 << 18, 0.3))
    assert not np.array_equal(k, T.dropout_keep(9, 5, 1, 1 << 18, 0.3))
    assert T.dropout_keep(1, 1, 1, 100, 0.0).all()


@pytest.mark.parametrize("tag", ["plain", "cpg", "cpg_mlp"])
def test_oracle_gradients_match_reference_sister_models_autograd(golden_dir, tag):
    """The training oracle's loss and gradients against torch autograd through the REFERENCE's own PyTorch forward
    (fact_network.py ConvE / CPG_ConvE) and loss (emb.py:50-58), eval-mode BN, no dropout (fixture:
    oracle/gen_golden.py gen_minerva_grads): pins the backward of the shared structure to reference code."""
    import os
    from oracle import coper_train_oracle as T
    from tests.minerva_map import load_grad_case, reference_grads_in_our_layout
    g = np.load(os.path.join(golden_dir, "minerva_grads.npz"))
    md, p, sd, batch = load_grad_case(g, tag)
    ref = {k: np.array(v, np.float64) for k, v in p.items()}
    opt = T.AMSGrad(T.trainable_names(md), ref, lr=md["learning_rate"])
    loss, grads, gn = T.train_step(ref, md, dict(e1=batch["e1"], rel=batch["rel"], lookup=None, labels=batch["labels"]), opt,
                                   seed=0, step=0, momentum=0.1)
    assert abs(loss - float(g[tag + ":loss"])) < 2e-6 * abs(loss)
    for leaf, (want, got) in reference_grads_in_our_layout(g, tag, sd, grads).items():
        want, got = np.asarray(want, np.float64), np.asarray(got, np.float64).reshape(np.shape(want))
        assert np.abs(got - want).max() < 2e-5 * max(np.abs(want).max(), 1e-6) + 1e-9, leaf


@pytest.mark.parametrize("tag", ["plain", "cpg", "cpg_mlp"])
def test_inference_oracle_matches_sister_model_scores_incl_mlp_generator(golden_dir, tag):
    """Inference forward of the oracle (generators evaluated per relation) against the sister models' own sigmoid
    scores stored with the gradient fixture -- adds the g_MLP + generator-BN variant to minerva_e2e's two."""
    import os
    from tests.minerva_map import load_grad_case, sigmoid
    g = np.load(os.path.join(golden_dir, "minerva_grads.npz"))
    md, p, sd, batch = load_grad_case(g, tag)
    out = O.forward({k: np.asarray(v, np.float32) for k, v in p.items()}, md, batch["e1"], batch["rel"], np.float64)
    s = out["h"] @ np.asarray(p["ent_emb"], np.float64).T + np.asarray(p["pred_bias"], np.float64)
    assert np.abs(sigmoid(s) - g[tag + ":S"]).max() < 2e-6
