#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  TEST INFRASTRUCTURE ONLY; runs ONLY in the authoring container
(it reads /root/reference, which does not exist on the GPU box).  The fixtures it writes are data
(inputs + expected outputs); no reference source is copied.

  rank_*.npz        inputs (pred, e2, CSR filter) -> outputs of the REFERENCE'S OWN
                    `ranking_and_hits` (CoPER_ConvE/qa_cpg/metrics.py:23-86), imported from
                    /root/reference under a stub `tensorflow` module that provides the only TF symbol
                    the file uses (`tf.errors.OutOfRangeError`, metrics.py:59), driven by a fake
                    session that feeds the batches.
  cpg_substeps.npz  generator / generated-dense / score sub-steps computed by the reference's
                    PyTorch sister implementation (CoPER_MINERVA/src/emb/fact_network.py:228-259,
                    376-387) on CPU.
  conv_torch.npz    conv + BN stage cross-check against torch.nn.functional (not reference code:
                    the TF-1.14 graph cannot run here -- this half of the oracle stays "unpinned").
  fwd_*.npz         seeded small models: all weights by leaf name, queries, and the oracle's stage
                    outputs in fp32 with an fp64 shadow -- what the HIP path is compared with.

Usage: python oracle/gen_golden.py   (from the repo root)
"""
from __future__ import annotations

import importlib.util
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from coper_amd import data as cdata  # noqa: E402
from oracle import coper_oracle as O  # noqa: E402


# ------------------------------------------------------------------------------------------------
def load_reference_metrics():
    class OutOfRangeError(Exception):
        pass

    tf = types.ModuleType("tensorflow")
    tf.errors = types.SimpleNamespace(OutOfRangeError=OutOfRangeError)
    sys.modules["tensorflow"] = tf
    spec = importlib.util.spec_from_file_location("ref_metrics", os.path.join(REF, "CoPER_ConvE/qa_cpg/metrics.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, OutOfRangeError


class FakeModel(object):
    e1, e2, rel, e2_multi, predictions_all, input_iterator_handle = "e1", "e2", "rel", "e2_multi", "pred", "handle"


class FakeSession(object):
    def __init__(self, batches, eod):
        self.it = iter(batches)
        self.eod = eod

    def run(self, fetches, feed_dict=None):
        try:
            b = next(self.it)
        except StopIteration:
            raise self.eod()
        return tuple(np.array(b[k], copy=True) for k in fetches)


def reference_ranking(ref, eod, pred, e2, e2_multi, batch=32):
    batches = []
    for s in range(0, len(pred), batch):
        sl = slice(s, s + batch)
        batches.append(dict(e1=np.zeros(len(pred[sl]), np.int64), e2=e2[sl], rel=np.zeros(len(pred[sl]), np.int64),
                            e2_multi=e2_multi[sl], pred=pred[sl]))
    with tempfile.TemporaryDirectory() as td:
        mr, mrr, hits = ref.ranking_and_hits(FakeModel(), td, "h", "golden", session=FakeSession(batches, eod))
    return float(mr), float(mrr), {int(k): float(v) for k, v in hits.items()}


def tie_free(rng, B, E):
    """Rows of pairwise-distinct float32 values: draw 2E, keep E distinct ones, shuffle."""
    pred = np.empty((B, E), np.float32)
    for i in range(B):
        u = np.unique((rng.standard_normal(2 * E) * 3).astype(np.float32))
        assert len(u) >= E
        pred[i] = rng.permutation(u)[:E]
    return pred


def gen_rank_fixtures(ref, eod):
    rng = np.random.default_rng(20240607)
    for E, B in ((14, 64), (257, 64), (4099, 16)):
        pred = tie_free(rng, B, E)
        e2 = rng.integers(0, E, B, dtype=np.int64)
        e2_multi = np.zeros((B, E), np.float32)
        for i in range(B):
            k = int(min(E - 1, rng.geometric(0.2)))
            e2_multi[i, rng.integers(0, E, k)] = 1.0
            e2_multi[i, e2[i]] = 1.0
        if E == 257:  # a few rows WITHOUT the target in the filter, and one empty filter
            e2_multi[0, e2[0]] = 0.0
            e2_multi[1, :] = 0.0
        mr, mrr, hits = reference_ranking(ref, eod, pred, e2, e2_multi)
        indptr, idx = O.dense_to_csr(e2_multi)
        ng, ne = O.rank_counts(pred, e2, indptr, idx)
        assert np.all(ne == 0)
        ks = sorted(hits)
        np.savez(os.path.join(OUT, "rank_E%d.npz" % E), pred=pred, e2=e2, filt_indptr=indptr, filt_idx=idx,
                 ref_mr=mr, ref_mrr=mrr, ref_hits_k=np.array(ks), ref_hits=np.array([hits[k] for k in ks]),
                 closed_form_rank=(1 + ng).astype(np.int64))
        # the closed form must reproduce the reference exactly on tie-free data
        m2 = O.metrics_from_ranks(1 + ng)
        assert m2[0] == mr and abs(m2[1] - mrr) < 1e-15, (m2, mr, mrr)
        print("rank_E%d: mr=%.6f mrr=%.6f hits=%s" % (E, mr, mrr, hits))
    # tie cases: the reference's rank under ties is argsort-order dependent; the fixture
    # records what the reference returned and the [lo, hi] band the closed form allows.
    E, B = 64, 8
    pred = np.zeros((B, E), np.float32)                      # all ties
    pred[4:] = np.round(rng.standard_normal((4, E)) * 2).astype(np.float32)   # heavy partial ties
    e2 = rng.integers(0, E, B, dtype=np.int64)
    e2_multi = np.zeros((B, E), np.float32)
    for i in range(B):
        e2_multi[i, e2[i]] = 1.0
        e2_multi[i, rng.integers(0, E, 3)] = 1.0
    ranks_ref = []
    for i in range(B):
        mr, _, _ = reference_ranking(ref, eod, pred[i:i + 1], e2[i:i + 1], e2_multi[i:i + 1])
        ranks_ref.append(int(round(mr)))
    indptr, idx = O.dense_to_csr(e2_multi)
    ng, ne = O.rank_counts(pred, e2, indptr, idx)
    ranks_ref = np.array(ranks_ref)
    assert np.all(ranks_ref >= 1 + ng) and np.all(ranks_ref <= 1 + ng + ne)
    np.savez(os.path.join(OUT, "rank_ties.npz"), pred=pred, e2=e2, filt_indptr=indptr, filt_idx=idx,
             ref_rank=ranks_ref, n_greater=ng, n_equal=ne)
    print("rank_ties: ref ranks", ranks_ref, "band lo", 1 + ng, "hi", 1 + ng + ne)


# ------------------------------------------------------------------------------------------------
def gen_cpg_substeps():
    import torch
    spec = importlib.util.spec_from_file_location("ref_fact_network", os.path.join(REF, "CoPER_MINERVA/src/emb/fact_network.py"))
    fn = importlib.util.module_from_spec(spec)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        spec.loader.exec_module(fn)
    torch.manual_seed(7)
    r, F, d, E, B = 20, 64, 40, 50, 12
    out = {}
    for tag, hidden in (("lin", []), ("mlp", [16])):
        with contextlib.redirect_stdout(io.StringIO()):
            gw = fn.ContextualParameterGenerator([r] + hidden, [F, d], dropout=0.2, use_batch_norm=False, use_bias=False)
            gb = fn.ContextualParameterGenerator([r] + hidden, [d], dropout=0.2, use_batch_norm=False, use_bias=False)
        gw.eval(); gb.eval()
        R = torch.randn(B, r) * 0.3
        X = torch.relu(torch.randn(B, F))
        E2 = torch.randn(E, d) * 0.3
        b = torch.randn(E) * 0.1
        with torch.no_grad():
            W = gw(R)                                              # fact_network.py:255-259
            fc = torch.einsum('ij, ijk-> ik', X, W.view(-1, F, d))  # fact_network.py:376-379
            fc = fc + gb(R)                                        # fact_network.py:380
            h = torch.relu(fc)
            S = torch.mm(h, E2.transpose(1, 0)) + b.expand(B, E)   # fact_network.py:386-387
        lin_w = [m.weight.detach().numpy().T.copy() for m in gw.network if isinstance(m, torch.nn.Linear)]
        lin_b = [m.weight.detach().numpy().T.copy() for m in gb.network if isinstance(m, torch.nn.Linear)]
        out.update({tag + "_R": R.numpy(), tag + "_X": X.numpy(), tag + "_E2": E2.numpy(), tag + "_b": b.numpy(),
                    tag + "_W": W.numpy(), tag + "_fc": fc.numpy(), tag + "_S": S.numpy(),
                    tag + "_nproj": np.array(len(lin_w))})
        for i, (a, c) in enumerate(zip(lin_w, lin_b)):
            out["%s_Pw%d" % (tag, i)] = a
            out["%s_Pb%d" % (tag, i)] = c
    np.savez(os.path.join(OUT, "cpg_substeps.npz"), **out)
    print("cpg_substeps: ok")


def gen_conv_torch():
    import torch
    import torch.nn.functional as Fn
    rng = np.random.default_rng(11)
    B, H, W, C = 5, 10, 20, 32
    img = rng.standard_normal((B, H, W)).astype(np.float32)
    K = rng.standard_normal((3, 3, 1, C)).astype(np.float32)
    kb = rng.standard_normal(C).astype(np.float32)
    bn = {k: v.astype(np.float32) for k, v in dict(gamma=rng.uniform(0.5, 1.5, C), beta=rng.standard_normal(C) * 0.1,
                                                    mean=rng.standard_normal(C) * 0.1, var=rng.uniform(0.5, 1.5, C)).items()}
    x = torch.from_numpy(img)[:, None]                                  # NCHW
    w = torch.from_numpy(K).permute(3, 2, 0, 1).contiguous()            # HWIO -> OIHW
    y = Fn.conv2d(x, w, torch.from_numpy(kb))
    y = Fn.batch_norm(y, torch.from_numpy(bn["mean"]), torch.from_numpy(bn["var"]), torch.from_numpy(bn["gamma"]),
                      torch.from_numpy(bn["beta"]), training=False, eps=1e-3)
    y = torch.relu(y).permute(0, 2, 3, 1).contiguous().numpy()          # NHWC
    np.savez(os.path.join(OUT, "conv_torch.npz"), img=img, K=K, kb=kb, out_nhwc=y, **{"bn_" + k: v for k, v in bn.items()})
    print("conv_torch: ok")


def gen_minerva_e2e():
    """End-to-end pin of conv -> dense -> score against the reference's PyTorch sister models
    (CoPER_MINERVA/src/emb/fact_network.py: ConvE :116-197, CPG_ConvE :261-439) run on CPU in eval mode with
    randomised BN statistics.  The fixture stores the reference-side tensors (torch layouts) and the
    reference outputs; tests/minerva_map.py maps them onto the qa_cpg leaf names."""
    import contextlib
    import io
    import torch
    spec = importlib.util.spec_from_file_location("ref_fact_network2", os.path.join(REF, "CoPER_MINERVA/src/emb/fact_network.py"))
    fn = importlib.util.module_from_spec(spec)
    with contextlib.redirect_stdout(io.StringIO()):
        spec.loader.exec_module(fn)
    torch.manual_seed(123)
    E, R, B, d1, d2, C = 60, 6, 10, 10, 4, 8
    d = d1 * d2
    out = {}

    class KG(object):
        def __init__(self, ent, rel):
            self.ent, self.rel = ent, rel
        def get_entity_embeddings(self, e):
            return self.ent[e]
        def get_relation_embeddings(self, r):
            return self.rel[r]
        def get_all_entity_embeddings(self):
            return self.ent

    def randomise_bn(bn):
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.1)
            bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.6, 1.4)

    for tag, r_dim, cpg in (("plain", d, False), ("cpg", 20, True)):
        args = types.SimpleNamespace(entity_dim=d, relation_dim=r_dim, emb_2D_d1=d1, emb_2D_d2=d2, num_out_channels=C,
                                     kernel_size=3, hidden_dropout_rate=0.3, feat_dropout_rate=0.2, cpg_conv_net=[-1],
                                     cpg_fc_net=[], cpg_dropout=0.2, cpg_batch_norm=False, cpg_batch_norm_momentum=0.1,
                                     cpg_use_bias=False)
        with contextlib.redirect_stdout(io.StringIO()):
            m = fn.CPG_ConvE(args, E) if cpg else fn.ConvE(args, E)
        m.eval()
        randomise_bn(m.bn0); randomise_bn(m.bn2)
        with torch.no_grad():
            m.b.normal_(0, 0.1)
            m.conv1.weight.normal_(0, 1.0); m.conv1.bias.normal_(0, 0.1)
            if cpg:
                for g_, std in ((m.fc_weights, 0.35), (m.fc_bias, 0.3)):
                    for lin in g_.network:
                        if isinstance(lin, torch.nn.Linear):
                            lin.weight.normal_(0, std)
            else:
                m.fc.weight.normal_(0, 0.12); m.fc.bias.normal_(0, 0.1)
        ent = torch.randn(E, d) * 0.3
        rel = torch.randn(R, r_dim) * 0.3
        kg = KG(ent, rel)
        e1 = torch.randint(0, E, (B,)); r = torch.randint(0, R, (B,)); e2 = torch.randint(0, E, (B,))
        with torch.no_grad():
            S = m.forward(e1, r, kg)                    # [B, E] sigmoid scores
            Sf = m.forward_fact(e1, r, e2, kg)          # [B, 1]
        out.update({tag + ":ent": ent.numpy(), tag + ":rel": rel.numpy(), tag + ":e1": e1.numpy(), tag + ":r": r.numpy(),
                    tag + ":e2": e2.numpy(), tag + ":S": S.numpy(), tag + ":S_fact": Sf.numpy(),
                    tag + ":dims": np.array([E, R, B, d1, d2, C, r_dim])})
        for k, v in m.state_dict().items():
            out[tag + ":sd:" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "minerva_e2e.npz"), **out)
    print("minerva_e2e: keys", len(out), "S range", float(out["plain:S"].min()), float(out["plain:S"].max()))


def gen_minerva_grads():
    """Training-side pin: loss and GRADIENTS of the reference's PyTorch sister models by torch autograd through the
    reference's own forward code (fact_network.py ConvE / CPG_ConvE .forward) and the reference's loss
    (`EmbeddingBasedMethod.loss`, src/emb/emb.py:50-58: label smoothing (1 - eps) * e2 + 1 / num_entities, BCELoss
    on the sigmoid scores, mean reduction), models in eval mode (BN on running statistics, no dropout).
    tests/ compare coper_train_step's loss and gradients against these through the tensor mapping of
    coper_amd.weights.from_minerva_state_dict (chain rule for the folded input BN)."""
    import contextlib
    import io
    import torch
    spec = importlib.util.spec_from_file_location("ref_fact_network3", os.path.join(REF, "CoPER_MINERVA/src/emb/fact_network.py"))
    fn = importlib.util.module_from_spec(spec)
    with contextlib.redirect_stdout(io.StringIO()):
        spec.loader.exec_module(fn)
    torch.manual_seed(321)
    E, R, B, d1, d2, C = 60, 6, 12, 10, 4, 8
    d = d1 * d2
    eps_ls = 0.1
    out = {}

    class KG(object):
        def __init__(self, ent, rel):
            self.ent, self.rel = ent, rel
        def get_entity_embeddings(self, e):
            return self.ent[e]
        def get_relation_embeddings(self, r):
            return self.rel[r]
        def get_all_entity_embeddings(self):
            return self.ent

    def randomise_bn(bn):
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.1)
            bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.6, 1.4)

    for tag, r_dim, cpg, fc_net, gen_bn in (("plain", d, False, [], False), ("cpg", 20, True, [], False),
                                            ("cpg_mlp", 20, True, [12], True)):
        args = types.SimpleNamespace(entity_dim=d, relation_dim=r_dim, emb_2D_d1=d1, emb_2D_d2=d2, num_out_channels=C,
                                     kernel_size=3, hidden_dropout_rate=0.3, feat_dropout_rate=0.2, cpg_conv_net=[-1],
                                     cpg_fc_net=fc_net, cpg_dropout=0.2, cpg_batch_norm=gen_bn, cpg_batch_norm_momentum=0.1,
                                     cpg_use_bias=False)
        with contextlib.redirect_stdout(io.StringIO()):
            m = fn.CPG_ConvE(args, E) if cpg else fn.ConvE(args, E)
        m.eval()
        randomise_bn(m.bn0); randomise_bn(m.bn2)
        with torch.no_grad():
            m.b.normal_(0, 0.1)
            m.conv1.weight.normal_(0, 1.0); m.conv1.bias.normal_(0, 0.1)
            if cpg:
                # (moderate logits: the sister's fp32 sigmoid + BCELoss saturates and clamps log(0) to -100 beyond |s| ~ 17,
                # which is an artefact of that formulation, not of the model)
                for g_, std in ((m.fc_weights, 0.35 if not fc_net else 0.12), (m.fc_bias, 0.3 if not fc_net else 0.15)):
                    for lin in g_.network:
                        if isinstance(lin, torch.nn.Linear):
                            lin.weight.normal_(0, std)
                        if isinstance(lin, torch.nn.BatchNorm1d):
                            randomise_bn(lin)
            else:
                m.fc.weight.normal_(0, 0.12); m.fc.bias.normal_(0, 0.1)
        ent = (torch.randn(E, d) * 0.3).requires_grad_(True)
        rel = (torch.randn(R, r_dim) * 0.3).requires_grad_(True)
        kg = KG(ent, rel)
        e1 = torch.randint(0, E, (B,)); r = torch.randint(0, R, (B,))
        labels = (torch.rand(B, E) < 0.08).float()
        labels[torch.arange(B), torch.randint(0, E, (B,))] = 1.0
        pred = m.forward(e1, r, kg)                                         # reference forward, sigmoid scores [B, E]
        e2_label = ((1 - eps_ls) * labels) + (1.0 / labels.size(1))         # emb.py:55
        loss = torch.nn.BCELoss()(pred, e2_label)                           # emb.py:35,57
        loss.backward()
        out.update({tag + ":ent": ent.detach().numpy(), tag + ":rel": rel.detach().numpy(), tag + ":e1": e1.numpy(),
                    tag + ":r": r.numpy(), tag + ":labels": labels.numpy(), tag + ":loss": np.float64(loss.item()),
                    tag + ":eps_ls": np.float64(eps_ls), tag + ":dims": np.array([E, R, B, d1, d2, C, r_dim]),
                    tag + ":grad:ent": ent.grad.numpy(), tag + ":grad:rel": rel.grad.numpy(),
                    tag + ":S": pred.detach().numpy()})
        for k, v in m.state_dict().items():
            out[tag + ":sd:" + k] = v.numpy()
        for k, v in m.named_parameters():
            if v.grad is not None:
                out[tag + ":grad:" + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "minerva_grads.npz"), **out)
    print("minerva_grads: keys", len(out), "losses", float(out["plain:loss"]), float(out["cpg:loss"]), float(out["cpg_mlp:loss"]))


def gen_loader_fixture():
    """TSV -> JSON -> id maps through the REFERENCE'S OWN loader code (qa_cpg/data.py load_and_preprocess,
    _write_graph, _assign_ids), imported under the stub tensorflow module, on a small split of the
    nell-995 dev triples the reference ships (CoPER_ConvE/data/nell-995-test/dev.txt)."""
    import shutil
    spec = importlib.util.spec_from_file_location("ref_data", os.path.join(REF, "CoPER_ConvE/qa_cpg/data.py"))
    rd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rd)
    lines = open(os.path.join(REF, "CoPER_ConvE/data/nell-995-test/dev.txt")).read().splitlines(True)
    tsv_dir = os.path.join(OUT, "kg_tsv")
    os.makedirs(tsv_dir, exist_ok=True)
    splits = {"train.txt": lines[:400], "dev.txt": lines[400:470], "test.txt": lines[470:]}
    for name, ls in splits.items():
        with open(os.path.join(tsv_dir, name), "w") as f:
            f.writelines(ls)
    for clean in (False, True):
        with tempfile.TemporaryDirectory() as td:
            for name in splits:
                shutil.copy(os.path.join(tsv_dir, name), td)
            loader = rd.NELL995Loader(is_test=True, needs_test_set_cleaning=clean)
            json_files = loader.load_and_preprocess(td)
            loader._assign_ids(json_files)
            dst = os.path.join(OUT, "kg_ref_clean" if clean else "kg_ref")
            os.makedirs(dst, exist_ok=True)
            for fn_ in list(json_files.values()) + [os.path.join(td, "entities.txt"), os.path.join(td, "relations.txt")]:
                shutil.copy(fn_, dst)
            print("loader fixture (clean=%s):" % clean, {k: sum(1 for _ in open(v)) for k, v in json_files.items()})


# ------------------------------------------------------------------------------------------------
FWD_CASES = {
    # name: (model_descriptors overrides, #queries)
    "plain": (dict(num_ent=257, num_rel=22, ent_emb_size=40, rel_emb_size=40, emb_h=10, emb_w=4, conv_num_channels=8,
                   context_rel_conv=None, context_rel_out=None), 48),
    "cpg_fc": (dict(num_ent=257, num_rel=22, ent_emb_size=40, rel_emb_size=8, emb_h=10, emb_w=4, conv_num_channels=8,
                    context_rel_conv=None, context_rel_out=[]), 48),
    "cpg_fc_mlp": (dict(num_ent=257, num_rel=22, ent_emb_size=40, rel_emb_size=8, emb_h=10, emb_w=4,
                        conv_num_channels=8, context_rel_conv=None, context_rel_out=[12]), 48),
    "cpg_conv_fc": (dict(num_ent=131, num_rel=10, ent_emb_size=32, rel_emb_size=8, emb_h=4, emb_w=8,
                         conv_num_channels=8, context_rel_conv=[], context_rel_out=[]), 40),
    "cpg_conv_only_concat": (dict(num_ent=131, num_rel=10, ent_emb_size=32, rel_emb_size=6, emb_h=4, emb_w=8,
                                  conv_num_channels=8, context_rel_conv=[5], context_rel_out=None, concat_rel=True,
                                  context_rel_use_batch_norm=False), 40),
    "lookup": (dict(num_ent=131, num_rel=10, ent_emb_size=32, rel_emb_size=8, emb_h=4, emb_w=8, conv_num_channels=8,
                    context_rel_conv=[], context_rel_out=[], do_parameter_lookup=True), 40),
}


def gen_fwd_fixtures():
    for name, (over, Q) in FWD_CASES.items():
        md = dict(cdata._COMMON)
        md.update(over)
        p = cdata.synthetic_params(md, seed=3)
        q = cdata.synthetic_queries(md, Q, seed=5, mean_filter=3.0, max_filter=16)
        st32 = O.forward(p, md, q["e1"], q["rel"], np.float32, materialise=True)
        st64 = O.forward(p, md, q["e1"], q["rel"], np.float64, materialise=True)
        lg32 = O.score_all(st32["h"], p["ent_emb"], p["pred_bias"])
        lg64 = O.score_all(st64["h"], p["ent_emb"], p["pred_bias"])
        ng, ne = O.rank_counts(lg64, q["e2"], q["filt_indptr"], q["filt_idx"])
        t64 = lg64[np.arange(Q), q["e2"]]
        gap = np.abs(lg64 - t64[:, None])
        gap[np.arange(Q), q["e2"]] = np.inf
        blob = {"param:" + k: v for k, v in p.items()}
        blob.update({"q:" + k: v for k, v in q.items()})
        blob.update({"md_keys": np.array(sorted(over)), "md_repr": np.array(repr(over))})
        for k in ("x", "z", "h"):
            blob["f32:" + k] = st32[k]
            blob["f64:" + k] = st64[k]
        blob["f32:logits"] = lg32
        blob["f64:logits"] = lg64
        blob["f64:n_greater"] = ng
        blob["f64:n_equal"] = ne
        blob["f64:min_gap"] = gap.min(axis=1)
        np.savez_compressed(os.path.join(OUT, "fwd_%s.npz" % name), **blob)
        print("fwd_%s: |logit| max %.3f  max|f32-f64| %.2e  min rank gap %.2e" % (
            name, np.abs(lg64).max(), np.abs(lg32 - lg64).max(), gap.min()))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if not os.path.isdir(REF):
        sys.exit("gen_golden.py needs /root/reference (authoring container only)")
    ref, eod = load_reference_metrics()
    gen_rank_fixtures(ref, eod)
    gen_cpg_substeps()
    gen_conv_torch()
    gen_minerva_e2e()
    gen_minerva_grads()
    gen_loader_fixture()
    gen_fwd_fixtures()
    print("sizes:", {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))})
