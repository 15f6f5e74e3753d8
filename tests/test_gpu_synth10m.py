"""GPU: BASELINE.json configs[4] at FULL size on one MI355X -- the 10M-entity KG (|E| = 10^7, R2 = 2000, d = 256),
4096 queries per pass: the fused ranks against recomputation from materialised logit rows, the logits against the
fp64 oracle on sampled (query, entity) pairs, two 5M-row shard handles against the unsharded handle (counts and merged
top-10), through the C ABI.  The table is drawn on the device in independently seeded row blocks
(coper_amd.data.synthetic_entity_rows_device), so every sharding sees the same KG."""
import numpy as np
import pytest
import torch

from coper_amd import data as cdata

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3
NAME, Q, K = "synth10m_cpg", 4096, 10


def _handle(md, small, ent, bias, mode, shard=None):
    from coper_amd.models import ConvE
    lo, hi = shard if shard else (0, md["num_ent"])
    m = ConvE(md, device="cuda:0", shard=shard, score_mode=mode)
    p = dict(small)
    p["ent_emb"], p["pred_bias"] = ent[lo:hi], bias[lo:hi]
    m.load_parameters(p, global_rows=False)
    return m.prepare()


def test_entity_table_does_not_depend_on_the_sharding():
    md = cdata.model_descriptors(NAME)
    lo, hi = 3 * cdata.ENTITY_SEED_BLOCK - 1000, 5 * cdata.ENTITY_SEED_BLOCK + 77
    ent, bias = cdata.synthetic_entity_rows_device(md, 0, "cuda:0", lo, hi)
    for a, b in ((lo, lo + 500), (4 * cdata.ENTITY_SEED_BLOCK - 3, 4 * cdata.ENTITY_SEED_BLOCK + 9), (hi - 100, hi)):
        e2, b2 = cdata.synthetic_entity_rows_device(md, 0, "cuda:0", a, b)
        assert torch.equal(e2, ent[a - lo:b - lo]) and torch.equal(b2, bias[a - lo:b - lo])
    assert abs(float(ent.std()) - 0.1) < 1e-3 and abs(float(bias.std()) - 0.1) < 1e-3


@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
def test_synth10m_full_size(oracle_chain, mode):
    O = oracle_chain
    from coper_amd.sharding import merge_topk
    md = cdata.model_descriptors(NAME)
    E, d = md["num_ent"], md["ent_emb_size"]
    small = cdata.synthetic_params(md, 0, skip=("ent_emb", "pred_bias"))
    ent, bias = cdata.synthetic_entity_rows_device(md, 0, "cuda:0")
    q = cdata.synthetic_queries(md, Q, seed=0)
    dq = {k: torch.as_tensor(v).cuda() for k, v in q.items()}
    full = _handle(md, small, ent, bias, mode)
    ranks, ne, h = full.rank_pass(dq["e1"], dq["rel"], dq["e2"], dq["filt_indptr"], dq["filt_idx"], want_h=True)
    tgt = full.target_scores(h, dq["e2"])
    ng_k, ne_k, tv, ti = full.rank_counts(h, tgt, dq["e2"], dq["filt_indptr"], dq["filt_idx"], k=K)
    assert torch.equal(1 + ng_k, ranks) and torch.equal(ne_k, ne)          # the top-k route counts the same
    ranks_np = ranks.cpu().numpy()
    assert ranks_np.min() >= 1 and ranks_np.max() <= E

    # (1) a query sample: the reference ranker's semantics (metrics.py:44-50) applied to the materialised logit rows
    sample = np.arange(5, Q, 113)                                            # 37 queries x 10M logits = 1.5 GB
    hs = h[torch.as_tensor(sample).cuda()].contiguous()
    rows = full.score_all(hs)
    # the logits that define the ranks: the fp32 chain on this h -- the mode's own in the f32 mode, an f32 handle's in the
    # bf16x3 mode (its count kernel hands every comparison closer than its error to that chain: the exact band)
    if mode == "f32":
        xrows = rows.clone()
    else:
        ref = _handle(md, small, ent, bias, "f32")
        xrows = ref.score_all(hs)
        ref.close()
        del ref
    ip, ix = q["filt_indptr"], q["filt_idx"]
    for i, b in enumerate(sample):
        row, xrow = rows[i], xrows[i]
        t, tx = row[q["e2"][b]].clone(), xrow[q["e2"][b]].clone()
        assert t.item() == tgt[0, b].item() and tx.item() == tgt[1, b].item()   # pair kernels == tile kernels
        filt = torch.as_tensor(ix[ip[b]:ip[b + 1]]).cuda()
        row[filt] = float("-inf")
        xrow[filt] = float("-inf")
        row[q["e2"][b]] = t
        xrow[q["e2"][b]] = tx
        assert int((xrow > tx).sum().item()) + 1 == ranks_np[b]
        assert int((xrow == tx).sum().item()) - 1 == int(ne[b].item())
        ev, ei = torch.topk(row, K)                                          # ties at the top are not expected here
        assert torch.equal(ev, tv[b]) and torch.equal(ei, ti[b])
    del xrows
    # (1b) round 6: the same ranks pinned to the C RESTATEMENT of the fp32 chain directly -- no second HIP handle
    # in the argument.  Eight of the sampled queries: their 10M logits by oracle_score_chain (oracle/coper_oracle_chain.c, fmaf in
    # the documented order) block-wise over the device-drawn table copied to the host, the closed form of metrics.py:44-50 on them.
    from concurrent.futures import ThreadPoolExecutor
    pin = sample[::5][:8]
    hp = np.ascontiguousarray(h[torch.as_tensor(pin).cuda()].cpu().numpy())
    tq = q["e2"][pin]
    tx_c = np.array([O.score_chain(hp[i:i + 1], ent[int(tq[i]):int(tq[i]) + 1].cpu().numpy(), bias[int(tq[i]):int(tq[i]) + 1].cpu().numpy())[0, 0]
                     for i in range(len(pin))], np.float32)
    ng_c, ne_c = np.zeros(len(pin), np.int64), np.zeros(len(pin), np.int64)
    BLK = 1_000_000
    with ThreadPoolExecutor(max_workers=8) as ex:
        for lo in range(0, E, BLK):
            hi = min(E, lo + BLK)
            Eb, bb = ent[lo:hi].cpu().numpy(), bias[lo:hi].cpu().numpy()

            def one(i):
                row = O.score_chain(hp[i:i + 1], Eb, bb)[0]                 # (ctypes releases the GIL: eight chains side by side)
                f = ix[ip[pin[i]]:ip[pin[i] + 1]]
                f = f[(f >= lo) & (f < hi)] - lo
                keep = np.ones(hi - lo, bool)
                keep[f] = False
                if lo <= tq[i] < hi:
                    keep[tq[i] - lo] = False
                row = row[keep]
                return int((row > tx_c[i]).sum()), int((row == tx_c[i]).sum())
            for i, (g_, e_) in enumerate(ex.map(one, range(len(pin)))):
                ng_c[i] += g_
                ne_c[i] += e_
    assert np.array_equal(tx_c, tgt[1, torch.as_tensor(pin).cuda()].cpu().numpy())            # the exact-chain targets, bit for bit
    assert np.array_equal(1 + ng_c, ranks_np[pin]) and np.array_equal(ne_c, ne.cpu().numpy()[pin]), (1 + ng_c, ranks_np[pin])
    # (2) logits on sampled columns against the fp64 oracle; fp32 mode: bit-equal to the documented chain
    cols = np.unique(np.concatenate([np.random.default_rng(1).integers(0, E, 3000), q["e2"][sample], q["e1"][sample]]))
    cols_t = torch.as_tensor(cols).cuda()
    p_s = dict(small)
    p_s["ent_emb"], p_s["pred_bias"] = ent[cols_t].cpu().numpy(), bias[cols_t].cpu().numpy()
    md_s = dict(md, num_ent=len(cols))
    pos = {int(c): j for j, c in enumerate(cols)}
    e1_s = np.array([pos[int(c)] for c in q["e1"][sample]])
    st = O.forward(p_s, md_s, e1_s, q["rel"][sample], np.float64, materialise=False)
    hn = hs.cpu().numpy()
    assert np.abs(hn - st["h"]).max() < 2e-4
    lg64 = O.score_all(st["h"], p_s["ent_emb"].astype(np.float64), p_s["pred_bias"].astype(np.float64))
    got = rows_sample = full.score_all(hs)[:, cols_t].cpu().numpy()
    err = float(np.abs(got - lg64).max())
    assert err < LOGIT_TOL and err < (3e-5 if mode == "f32" else 4e-4), err
    if mode == "f32":
        assert np.array_equal(got, O.score_chain(hn, p_s["ent_emb"], p_s["pred_bias"]))
    del rows, rows_sample, got

    # (3) two 5M-row shard handles == the unsharded handle: gathered rows, targets, counts, merged top-10
    shards = [_handle(md, small, ent, bias, mode, (0, E // 2)), _handle(md, small, ent, bias, mode, (E // 2, E))]
    e1_rows = sum(s.gather_entities(dq["e1"]) for s in shards)
    assert torch.equal(e1_rows, ent[dq["e1"]])
    h2 = shards[1].encode(dq["e1"], dq["rel"], e1_rows=e1_rows)
    assert torch.equal(h2, h)
    tgt2 = sum(s.target_scores(h2, dq["e2"]) for s in shards)
    assert torch.equal(tgt2, tgt)
    parts = [s.rank_counts(h2, tgt2, dq["e2"], dq["filt_indptr"], dq["filt_idx"], k=K) for s in shards]
    assert torch.equal(1 + parts[0][0] + parts[1][0], ranks) and torch.equal(parts[0][1] + parts[1][1], ne)
    mv, mi = merge_topk(torch.cat([x[2] for x in parts], 1), torch.cat([x[3] for x in parts], 1), K)
    assert torch.equal(mv, tv) and torch.equal(mi, ti)
    for x in shards + [full]:
        x.close()
