"""Seeded random model shapes through the whole path: sizes the BASELINE configs do not have --
embedding sizes that are not multiples of 8 or 16, few or many channels, other filter sizes, tiny entity tables, single-query
batches -- in both arithmetic modes.  Every case is checked against the fp64 oracle (h, logits) and for the consistency the
kernels promise among themselves: the fused ranks equal the closed form on the library's own logits bit for bit, the
pruned top-k equals the masked row top-k."""
import numpy as np
import pytest
import torch

from coper_amd import data as cdata
from tests.helpers import rank_defining_logits

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3      # north_star: "logits within 1e-3 fp32"
H_TOL = 2e-4


def _random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    while True:
        emb_h = int(rng.integers(3, 17))
        emb_w = int(rng.integers(3, 25))
        d = emb_h * emb_w
        if d <= 320:      # coper_create: the 128-query tile of the count kernels has to fit in LDS
            break
    variant = ["cpg_fc", "cpg_conv_fc", "cpg_mlp", "lookup", "plain", "cpg_fc_concat"][seed % 6]
    fh, fw = [(3, 3), (3, 3), (2, 2), (1, 3), (3, 2)][int(rng.integers(0, 5))]
    C = int(rng.choice([4, 8, 16, 32, 32, 40]))
    md = dict(cdata._COMMON)
    md.update(num_ent=int(rng.choice([5, 31, 33, 64, 257, 700])), num_rel=int(rng.choice([2, 6, 22, 40])),
              ent_emb_size=d, rel_emb_size=int(rng.choice([1, 3, 8, 10, 37, 50])), emb_h=emb_h, emb_w=emb_w,   # (config_*: 1, 8, 32, 37, 50, 200)
              conv_filter_height=fh, conv_filter_width=fw, conv_num_channels=C)
    if variant == "cpg_fc":
        md.update(context_rel_conv=None, context_rel_out=[])
    elif variant == "cpg_conv_fc":
        md.update(context_rel_conv=[], context_rel_out=[])
    elif variant == "cpg_mlp":
        md.update(context_rel_conv=[5], context_rel_out=[7, 6])
    elif variant == "lookup":
        md.update(context_rel_conv=[], context_rel_out=[], do_parameter_lookup=True)
    elif variant == "plain":    # the relation row is stacked under the entity image: same width, r = d
        md.update(context_rel_conv=None, context_rel_out=None, rel_emb_size=d)
    else:
        md.update(context_rel_conv=None, context_rel_out=[], concat_rel=True)
    Q = int(rng.choice([1, 31, 33, 129, 300, 520]))
    return variant, md, Q


import os as _os


@pytest.mark.parametrize("seed", range(int(_os.environ.get("COPER_EVAL_FUZZ", "60"))))      # (COPER_EVAL_FUZZ=600: the same test as a soak)
def test_random_shapes_against_oracle(oracle_chain, seed):
    from coper_amd.models import ConvE
    O = oracle_chain
    variant, md, Q = _random_case(seed)
    p = cdata.synthetic_params(md, seed=seed)
    q = cdata.synthetic_queries(md, Q, seed=seed)
    E = md["num_ent"]
    st = O.forward(p, md, q["e1"], q["rel"], np.float64, materialise=False)
    ref_logits = O.score_all(st["h"], p["ent_emb"].astype(np.float64), p["pred_bias"].astype(np.float64))
    mask = cdata.csr_to_dense_filter(q["filt_indptr"], q["filt_idx"], E).astype(bool)
    for mode in ("f32", "bf16x3"):
        m = ConvE(md, device="cuda:0", score_mode=mode)
        m.load_parameters(p)
        m.prepare()
        h = m.encode(q["e1"], q["rel"])
        assert np.abs(h.cpu().numpy() - st["h"]).max() < H_TOL, (variant, md, mode)
        logits = m.score_all(h).cpu().numpy()
        assert np.abs(logits - ref_logits).max() < LOGIT_TOL, (variant, md, mode)
        # the fused pass (no logits) against the closed form on the library's own logits: the count kernels, the pair
        # kernels and score_all compute the same bits
        ranks, ne = m.rank_pass(q["e1"], q["rel"], q["e2"], q["filt_indptr"], q["filt_idx"])
        xl = rank_defining_logits(O, m, h, p)      # bf16x3: the fp32 chain on this h (exact band), not the mode's own logits
        tgt = xl[np.arange(Q), q["e2"]]
        keep = ~mask
        keep[np.arange(Q), q["e2"]] = False
        want = 1 + ((xl > tgt[:, None]) & keep).sum(axis=1)
        want_eq = ((xl == tgt[:, None]) & keep).sum(axis=1)
        assert np.array_equal(ranks.cpu().numpy(), want), (variant, md, mode)
        assert np.array_equal(ne.cpu().numpy(), want_eq), (variant, md, mode)
        r0, _ = m.rank_pass(q["e1"], q["rel"], q["e2"], q["filt_indptr"], q["filt_idx"], want_equal=False)
        assert torch.equal(r0, ranks)
        # top-k of the filtered rows (known answers except the target masked), (score desc, id asc)
        k = min(5, E)
        out = m.rank_counts(h, m.target_scores(h, q["e2"]), q["e2"], q["filt_indptr"], q["filt_idx"], k=k)
        masked = np.where(keep | (np.arange(E)[None, :] == q["e2"][:, None]), logits, -np.inf)
        order = np.lexsort((np.broadcast_to(np.arange(E), masked.shape), -masked), axis=1)[:, :k]
        want_val = np.take_along_axis(masked, order, axis=1)
        got_val, got_idx = out[2].cpu().numpy(), out[3].cpu().numpy()
        assert np.array_equal(got_val, want_val), (variant, md, mode)
        finite = np.isfinite(want_val)
        assert np.array_equal(got_idx[finite], order[finite]), (variant, md, mode)
        m.close()


@pytest.mark.parametrize("seed", range(16))
def test_random_shards_sum_to_the_unsharded_result(seed):
    """SURVEY.md 8(e) on random shapes: the entity table cut at random rows into 2 or 3 shard handles on one GPU, the exchange
    done by hand -- gathered e1 rows, targets and rank counts add up to the unsharded handle's, the concatenated logits and
    the merged top-k are the unsharded ones, bit for bit, in both modes."""
    from coper_amd.models import ConvE
    from coper_amd.sharding import merge_topk
    variant, md, Q = _random_case(100 + seed)
    rng = np.random.default_rng(7000 + seed)
    E = md["num_ent"]
    if E < 8:
        md["num_ent"] = E = 40
    p = cdata.synthetic_params(md, seed=seed)
    q = cdata.synthetic_queries(md, Q, seed=seed)
    cuts = sorted(set(int(c) for c in rng.integers(1, E, size=int(rng.integers(1, 3)))))
    bounds = list(zip([0] + cuts, cuts + [E]))
    mode = "bf16x3" if seed % 2 else "f32"

    def make(shard=None):
        m = ConvE(md, device="cuda:0", score_mode=mode, shard=shard)
        m.load_parameters(p)
        m.prepare()
        return m

    full = make()
    h = full.encode(q["e1"], q["rel"])
    tgt_full = full.target_scores(h, q["e2"])
    k = min(4, min(hi - lo for lo, hi in bounds))
    ng_f, ne_f, tv_f, ti_f = full.rank_counts(h, tgt_full, q["e2"], q["filt_indptr"], q["filt_idx"], k=k)
    shards = [make(b) for b in bounds]
    rows = sum(s.gather_entities(q["e1"]) for s in shards)
    assert np.array_equal(rows.cpu().numpy(), np.asarray(p["ent_emb"], np.float32)[q["e1"]])
    hs = shards[-1].encode(q["e1"], q["rel"], e1_rows=rows)
    assert torch.equal(hs, h), (variant, md, bounds, mode)
    tgt = sum(s.target_scores(hs, q["e2"]) for s in shards)
    assert torch.equal(tgt, tgt_full)
    outs = [s.rank_counts(hs, tgt, q["e2"], q["filt_indptr"], q["filt_idx"], k=k) for s in shards]
    assert torch.equal(sum(o[0] for o in outs), ng_f) and torch.equal(sum(o[1] for o in outs), ne_f), (variant, md, bounds, mode)
    tv, ti = merge_topk(torch.cat([o[2] for o in outs], dim=1), torch.cat([o[3] for o in outs], dim=1), k)
    assert torch.equal(tv, tv_f), (variant, md, bounds, mode)
    fin = torch.isfinite(tv_f)
    assert torch.equal(ti[fin], ti_f[fin]), (variant, md, bounds, mode)
    assert torch.equal(torch.cat([s.score_all(hs) for s in shards], dim=1), full.score_all(h))
    for s in shards + [full]:
        s.close()


@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
def test_one_handle_many_calls_of_changing_size(oracle_chain, mode):
    """State that outlives a call -- the two relation-count buffers that zero each other, cursors, tickets, workspaces that
    grow, pooled timer events -- under 120 consecutive passes whose batch size jumps between 1 and 5,000 queries (single-launch
    and multi-launch grouping, one or many tiles, workspace regrowth), with the top-k route in between: every pass must equal
    the closed form on the handle's own logits."""
    from coper_amd.models import ConvE
    O = oracle_chain
    md = cdata.model_descriptors("fb15k237_cpg", num_ent=1500, num_rel=40)
    p = cdata.synthetic_params(md, 3)
    m = ConvE(md, device="cuda:0", score_mode=mode)
    m.load_parameters(p)
    m.prepare()
    E = md["num_ent"]
    rng = np.random.default_rng(99)
    sizes = [1, 5000, 33, 4096, 4097, 2, 512, 31, 129, 3000] + [int(x) for x in rng.integers(1, 5001, 110)]
    m.profile(True)
    for it, Q in enumerate(sizes):
        q = cdata.synthetic_queries(md, Q, seed=1000 + it)
        ranks, ne = m.rank_pass(q["e1"], q["rel"], q["e2"], q["filt_indptr"], q["filt_idx"])
        if it % 7 == 3:     # the top-k route reuses the count buffers and its own workspaces
            h = m.encode(q["e1"], q["rel"])
            m.rank_counts(h, m.target_scores(h, q["e2"]), q["e2"], q["filt_indptr"], q["filt_idx"], k=10)
        if it % 5 == 0 or Q < 64:
            h = m.encode(q["e1"], q["rel"])
            logits = rank_defining_logits(O, m, h, p)
            mask = cdata.csr_to_dense_filter(q["filt_indptr"], q["filt_idx"], E).astype(bool)
            tgt = logits[np.arange(Q), q["e2"]]
            keep = ~mask
            keep[np.arange(Q), q["e2"]] = False
            want = 1 + ((logits > tgt[:, None]) & keep).sum(axis=1)
            assert np.array_equal(ranks.cpu().numpy(), want), (it, Q)
        else:               # against the two-call path (its own grouping pass, other workspaces)
            h = m.encode(q["e1"], q["rel"])
            r2, ne2 = m.rank(h, q["e2"], q["filt_indptr"], q["filt_idx"])
            assert torch.equal(ranks, r2) and torch.equal(ne, ne2), (it, Q)
    assert m.check_ids() == 0 if hasattr(m, "check_ids") else True
    m.close()


def test_reference_style_dense_mask_batches():
    """The reference's batch contract carries a dense 0/1 mask e2_multi [B, |E|] (data.py:182-186, models.py:139-152).  Batches
    in that form go through ranking_and_hits like CSR batches: the mask is scanned on the model's device and the result is
    the CSR the loaders would have produced (same ranks)."""
    from coper_amd.data import dense_filter_to_csr
    from coper_amd.metrics import ranking_and_hits
    from coper_amd.models import ConvE
    md = cdata.model_descriptors("fb15k237_cpg", num_ent=3000, num_rel=30)
    m = ConvE(md, device="cuda:0", score_mode="bf16x3").load_parameters(cdata.synthetic_params(md, 1)).prepare()
    q = cdata.synthetic_queries(md, 700, seed=2)
    E = md["num_ent"]
    mask = cdata.csr_to_dense_filter(q["filt_indptr"], q["filt_idx"], E)
    ip_d, ix_d = dense_filter_to_csr(mask, device=m.device)
    ip_h, ix_h = dense_filter_to_csr(mask)
    assert np.array_equal(ip_d, ip_h) and np.array_equal(ix_d, ix_h)
    assert np.array_equal(ip_d, q["filt_indptr"]) and np.array_equal(ix_d, q["filt_idx"])
    csr_batches = [dict(e1=q["e1"][s:s + 128], e2=q["e2"][s:s + 128], rel=q["rel"][s:s + 128],
                        filt_indptr=q["filt_indptr"][s:s + 129] - q["filt_indptr"][s],
                        filt_idx=q["filt_idx"][q["filt_indptr"][s]:q["filt_indptr"][min(s + 128, 700)]]) for s in range(0, 700, 128)]
    dense_batches = [dict(e1=q["e1"][s:s + 128], e2=q["e2"][s:s + 128], rel=q["rel"][s:s + 128], e2_multi=mask[s:s + 128],
                          lookup_values=np.zeros((len(q["e1"][s:s + 128]), 0), np.int32)) for s in range(0, 700, 128)]
    a = ranking_and_hits(m, None, iter(csr_batches), "csr", return_ranks=True)
    b = ranking_and_hits(m, None, iter(dense_batches), "dense", return_ranks=True)
    assert a[:3] == b[:3] and np.array_equal(a[3], b[3])
    m.close()


@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
def test_filter_rows_with_thousands_of_known_answers(oracle_chain, mode):
    """A query whose filter list holds most of the entity table (real KGs have such (e1, rel) pairs): coper_encode_rank
    with and without tie counts (ranks only = the fused tail kernel, whose workgroup keeps the first 352 entries of its
    block and lists the block for k_filter_excess_bf16x3) and the evaluation loop give the closed-form ranks."""
    from coper_amd.metrics import ranking_and_hits
    from coper_amd.models import ConvE
    md = cdata.model_descriptors("fb15k237_cpg", num_ent=6000, num_rel=20)
    m = ConvE(md, device="cuda:0", score_mode=mode).load_parameters(cdata.synthetic_params(md, 2)).prepare()
    Q, E = 200, md["num_ent"]
    q = cdata.synthetic_queries(md, Q, seed=5)
    rng = np.random.default_rng(8)
    rows = [q["filt_idx"][q["filt_indptr"][i]:q["filt_indptr"][i + 1]] for i in range(Q)]
    for i, n in ((7, 5000), (8, 1500), (150, 3000)):
        rows[i] = np.unique(np.concatenate([rows[i], rng.choice(E, n, replace=False)]))
    indptr = np.zeros(Q + 1, np.int64)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    idx = np.concatenate(rows)
    h = m.encode(q["e1"], q["rel"])
    logits = rank_defining_logits(oracle_chain, m, h, cdata.synthetic_params(md, 2))
    mask = cdata.csr_to_dense_filter(indptr, idx, E).astype(bool)
    tgt = logits[np.arange(Q), q["e2"]]
    keep = ~mask
    keep[np.arange(Q), q["e2"]] = False
    want = 1 + ((logits > tgt[:, None]) & keep).sum(axis=1)
    fused, _ = m.rank_pass(q["e1"], q["rel"], q["e2"], indptr, idx)
    assert np.array_equal(fused.cpu().numpy(), want)
    for _ in range(3):       # (the excess list empties itself: every pass starts from zero)
        only, none = m.rank_pass(q["e1"], q["rel"], q["e2"], indptr, idx, want_equal=False)
        assert none is None and np.array_equal(only.cpu().numpy(), want)
    batches = [dict(e1=q["e1"], e2=q["e2"], rel=q["rel"], filt_indptr=indptr, filt_idx=idx)]
    out = ranking_and_hits(m, None, iter(batches), "heavy", return_ranks=True)
    assert np.array_equal(out[3], want)
    m.close()
