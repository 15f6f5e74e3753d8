#!/usr/bin/env python3
"""Randomised soak of the fused pass (coper_encode_rank, x3 mode) against the fp32 chain on the same embeddings:
    python tests/soak.py [cases] [seed]
Every case draws a model shape (d = 200 or 256, entity counts from a handful to tens of thousands, 2 - 600 relations, table
scale from 1e-4 to 10), a batch (1 - 6,000 queries, relation skew, duplicate queries) and filters (empty to thousands of known
answers in one row, duplicates of the target).  Checked per case:
  * ranks of the fused pass == ranks (and tie counts) of the fp32-exact ranker fed the pass's own h rows, for EVERY query;
  * the two-call path (encode + rank) gives the same ranks, and the same h bit for bit;
  * the band audit stays below 0.5;
  * the pruned top-k of the filtered rows (k drawn from 1 - 128; logits never materialised by the library) == a stable sort of
    the masked rows of the mode's own score_all logits: ids and values, (score desc, id asc), for every query (cases whose
    logit matrix stays under 2^26 elements).
  * every fourth case: the entity table cut at random rows into 2 - 3 shard handles, the exchange done by hand --
    gathered e1 rows, summed targets, summed counts, merged top-k -- equal to the unsharded handle's ranks, tie counts and top-k.
  * on a sample of 24 queries per case: the fp32-exact ranker's logits == the C restatement of the chain (oracle/coper_oracle_chain.c)
    bit for bit, and its ranks / tie counts == the C closed form of metrics.py:44-50 on them -- the checker itself is pinned to
    the CPU oracle in every case (test infrastructure: this file lives under tests/ because it loads the oracle).
Prints one line per case and a summary; exits non-zero on the first mismatch."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from coper_amd import data as cdata
from coper_amd.models import ConvE
from oracle import coper_oracle as O


def pinned_to_c_chain(p, h, q, m32, r32, ne32, n_sample=24):
    """f32-mode logits / ranks / tie counts of a query sample == the C oracle's on the same h rows."""
    Q = h.shape[0]
    sel = np.unique(np.linspace(0, Q - 1, min(Q, n_sample)).astype(np.int64))
    hs = h[torch.as_tensor(sel, device=h.device)].contiguous()
    chain = O.score_chain(hs.cpu().numpy(), p["ent_emb"], p["pred_bias"])
    if not np.array_equal(m32.score_all(hs).cpu().numpy(), chain):
        return False
    ip, ix = q["filt_indptr"], q["filt_idx"]
    sip = np.zeros(len(sel) + 1, np.int64)
    sip[1:] = np.cumsum(ip[sel + 1] - ip[sel])
    six = np.concatenate([ix[ip[i]:ip[i + 1]] for i in sel]).astype(np.int64)
    ng_c, ne_c = O.rank_counts_c(chain, q["e2"][sel], sip, six)
    return bool(np.array_equal(r32.cpu().numpy()[sel], 1 + ng_c) and np.array_equal(ne32.cpu().numpy()[sel], ne_c))


def draw_case(rng):
    base = "fb15k237_cpg" if rng.random() < 0.6 else "synth10m_cpg"
    E = int(rng.choice([7, 33, 200, 1000, 4097, 14541, 30000]))
    R = int(rng.choice([2, 6, 22, 110, 474, 600]))
    md = cdata.model_descriptors(base, num_ent=E, num_rel=R)
    ent_std = float(10.0 ** rng.uniform(-4, 1))
    Q = int(rng.choice([1, 2, 31, 33, 127, 129, 512, 1000, 4096, 4097, 6000]))
    return md, ent_std, Q


def draw_queries(md, Q, rng):
    E, Rf = int(md["num_ent"]), max(1, int(md["num_rel"]) // 2)
    kind = rng.integers(0, 4)
    rel = rng.integers(0, Rf, Q) if kind != 1 else np.minimum(rng.zipf(1.5, Q) - 1, Rf - 1)          # (skewed: a few relations carry the batch)
    e1 = rng.integers(0, E, Q)
    e2 = rng.integers(0, E, Q)
    if kind == 2 and Q > 4:                                                                        # duplicate queries
        e1[Q // 2:] = e1[:Q - Q // 2]; rel[Q // 2:] = rel[:Q - Q // 2]
    mean_f = float(rng.choice([0.0, 1.0, 4.0, 40.0]))
    rows = []
    for i in range(Q):
        n = 0 if mean_f == 0 else int(min(rng.geometric(1.0 / (1.0 + mean_f)) - 1, E))
        row = rng.integers(0, E, n)
        if rng.random() < 0.8:
            row = np.concatenate([row, e2[i:i + 1]])
        rows.append(np.unique(row))
    if Q > 3 and rng.random() < 0.3:                                                               # one row with most of the table known
        rows[int(rng.integers(0, Q))] = np.unique(rng.integers(0, E, min(E, 5000)))
    indptr = np.zeros(Q + 1, np.int64)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    idx = np.concatenate(rows).astype(np.int64) if indptr[-1] else np.zeros(0, np.int64)
    return dict(e1=e1.astype(np.int64), rel=rel.astype(np.int64), e2=e2.astype(np.int64), filt_indptr=indptr, filt_idx=idx)


def main():
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle")], stdout=subprocess.DEVNULL)
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    worst = 0.0
    t0 = time.time()
    for c in range(cases):
        md, ent_std, Q = draw_case(rng)
        p = cdata.synthetic_params(md, int(rng.integers(0, 1 << 30)), ent_std=ent_std)
        q = draw_queries(md, Q, rng)
        m = ConvE(md, device="cuda:0", score_mode="bf16x3", band_audit_period=1).load_parameters(p).prepare()
        m32 = ConvE(md, device="cuda:0", score_mode="f32").load_parameters(p).prepare()
        ranks, _, h = m.rank_pass(q["e1"], q["rel"], q["e2"], q["filt_indptr"], q["filt_idx"], want_equal=False, want_h=True)
        ranks_e, ne_e, h_e = m.rank_pass(q["e1"], q["rel"], q["e2"], q["filt_indptr"], q["filt_idx"], want_equal=True, want_h=True)
        r32, ne32 = m32.rank(h, q["e2"], q["filt_indptr"], q["filt_idx"], want_equal=True)
        pin_ok = pinned_to_c_chain(p, h, q, m32, r32, ne32)
        h2 = m.encode(q["e1"], q["rel"])
        r2, ne2 = m.rank(h2, q["e2"], q["filt_indptr"], q["filt_idx"], want_equal=True)
        topk_ok = True
        E = int(md["num_ent"])
        if Q * E <= (1 << 26):
            k = int(rng.choice([1, 3, 10, 32, 33, 128]))
            tgt = m.target_scores(h2, q["e2"])
            ng_k, ne_k, tv, ti = m.rank_counts(h2, tgt, q["e2"], q["filt_indptr"], q["filt_idx"], k=k)
            lg = m.score_all(h2)
            dq_e2 = torch.as_tensor(q["e2"], device=lg.device)
            ip = torch.as_tensor(q["filt_indptr"], device=lg.device)
            rows = torch.repeat_interleave(torch.arange(Q, device=lg.device), ip[1:] - ip[:-1])
            t = lg[torch.arange(Q, device=lg.device), dq_e2].clone()
            masked = lg.clone()
            if len(q["filt_idx"]):
                masked[rows, torch.as_tensor(q["filt_idx"], device=lg.device)] = float("-inf")
            masked[torch.arange(Q, device=lg.device), dq_e2] = t
            sv, si = torch.sort(masked, dim=1, descending=True, stable=True)
            kk = min(k, E)
            ev = torch.full((Q, k), float("-inf"), device=lg.device); ei = torch.full((Q, k), -1, dtype=torch.int64, device=lg.device)
            ev[:, :kk] = sv[:, :kk]; ei[:, :kk] = si[:, :kk]
            ei[ev == float("-inf")] = -1
            topk_ok = torch.equal(ti.to(torch.int64), ei) and torch.equal(tv, ev) and torch.equal(ng_k + 1, r32)
        shards_ok = True
        if c % 4 == 3 and E >= 8:
            from coper_amd.sharding import merge_topk
            cuts = sorted(set(int(x) for x in rng.integers(1, E, size=int(rng.integers(1, 3)))))
            bounds = list(zip([0] + cuts, cuts + [E]))
            ks = max(1, min(4, min(hi - lo for lo, hi in bounds)))
            shs = [ConvE(md, device="cuda:0", score_mode="bf16x3", shard=b, band_audit_period=1).load_parameters(p).prepare() for b in bounds]
            rows = sum(s_.gather_entities(q["e1"]) for s_ in shs)
            hs = shs[-1].encode(q["e1"], q["rel"], e1_rows=rows)
            tg = sum(s_.target_scores(hs, q["e2"]) for s_ in shs)
            outs = [s_.rank_counts(hs, tg, q["e2"], q["filt_indptr"], q["filt_idx"], k=ks) for s_ in shs]
            ng_f, ne_f, tv_f, ti_f = m.rank_counts(h2, m.target_scores(h2, q["e2"]), q["e2"], q["filt_indptr"], q["filt_idx"], k=ks)
            tv_s, ti_s = merge_topk(torch.cat([o[2] for o in outs], dim=1), torch.cat([o[3] for o in outs], dim=1), ks)
            fin = torch.isfinite(tv_f)
            shards_ok = (torch.equal(hs, h2) and torch.equal(sum(o[0] for o in outs), ng_f) and torch.equal(sum(o[1] for o in outs), ne_f)
                         and torch.equal(sum(o[0] for o in outs) + 1, r32) and torch.equal(tv_s, tv_f) and torch.equal(ti_s[fin], ti_f[fin])
                         and all(s_.band_audit()[0] < 0.5 for s_ in shs))
            for s_ in shs:
                s_.close()
        torch.cuda.synchronize()
        ratio, pairs = m.band_audit()
        ok = (torch.equal(ranks, r32) and torch.equal(ranks_e, r32) and torch.equal(ne_e, ne32) and torch.equal(r2, r32) and torch.equal(ne2, ne32)
              and torch.equal(h, h2) and torch.equal(h, h_e) and bool(torch.isfinite(h).all()) and ratio < 0.5 and topk_ok and shards_ok and pin_ok)
        worst = max(worst, ratio)
        print("case %3d  %-13s E=%-6d R=%-4d std=%-8.2g Q=%-5d nnz=%-7d  audit %.3f over %d pairs  %s" % (
            c, "d=%d" % md["ent_emb_size"], md["num_ent"], md["num_rel"], ent_std, Q, int(q["filt_indptr"][-1]), ratio, pairs, "ok" if ok else "MISMATCH"), flush=True)
        if not ok:
            bad = (ranks != r32).nonzero().flatten()[:8].tolist()
            print("   first differing queries:", bad, "fused", ranks[bad].tolist(), "chain", r32[bad].tolist(), "top-k ok:", topk_ok, "shards ok:", shards_ok, "C-oracle pin ok:", pin_ok)
            sys.exit(1)
        m.close(); m32.close()
    print("soak: %d cases, all ranks == the fp32 chain's on the same h (the chain pinned to the C oracle on a sample per case); largest band audit %.3f; %.0f s" % (cases, worst, time.time() - t0))


if __name__ == "__main__":
    main()
