"""TEST-ONLY scorer: the shard-scorer protocol of coper_amd.sharding implemented with the CPU
oracle, so the multi-rank exchange logic can run under gloo without a GPU."""
import numpy as np
import torch

from oracle import coper_oracle as O


class OracleShardScorer(object):
    device = torch.device("cpu")

    def __init__(self, params, md, shard):
        self.p, self.md, (self.lo, self.hi) = params, md, shard
        self.E = np.ascontiguousarray(params["ent_emb"][self.lo:self.hi])
        self.b = np.ascontiguousarray(params["pred_bias"][self.lo:self.hi])
        self.d = self.E.shape[1]
        self.num_ent = int(md["num_ent"])          # (lets the ranker split step 1 by owner)
        self._x3_absmax = None                     # the table-wide maximum the ranker agreed on (recorded, not used: fp32 arithmetic)
        self.absmax_calls = []

    def ent_absmax(self):
        return float(np.abs(self.E).max()) if self.E.size else 0.0

    def set_x3_ent_absmax(self, v):
        self._x3_absmax = float(v)
        self.absmax_calls.append(float(v))

    def owned_rows(self, ids):
        ids = np.asarray(ids)
        assert ((ids >= self.lo) & (ids < self.hi)).all(), "step 1 asked a shard for rows it does not hold"
        return torch.from_numpy(np.ascontiguousarray(self.E[ids - self.lo])), torch.from_numpy(np.ascontiguousarray(self.b[ids - self.lo]))

    def gather_entities(self, ids):
        ids = np.asarray(ids)
        out = np.zeros((len(ids), self.d), np.float32)
        own = (ids >= self.lo) & (ids < self.hi)
        out[own] = self.E[ids[own] - self.lo]
        return torch.from_numpy(out)

    def encode(self, e1, rel, e1_rows=None):
        rel = np.asarray(rel)
        p = dict(self.p)
        if e1_rows is not None:
            p["ent_emb"] = e1_rows.numpy()
            e1 = np.arange(len(rel))
        st = O.forward(p, self.md, e1, rel, np.float32, materialise=False)
        return torch.from_numpy(np.ascontiguousarray(st["h"]))

    def gather_bias(self, ids):
        ids = np.asarray(ids)
        out = np.zeros(len(ids), np.float32)
        own = (ids >= self.lo) & (ids < self.hi)
        out[own] = self.b[ids[own] - self.lo]
        return torch.from_numpy(out)

    def score_rows(self, h, rows, bias):
        h, rows, bias = h.numpy(), rows.numpy(), bias.numpy()
        out = np.array([O.score_chain(h[b:b + 1], rows[b:b + 1], bias[b:b + 1])[0, 0] for b in range(len(h))], np.float32)
        return torch.from_numpy(out)

    def target_scores(self, h, e2):
        e2 = np.asarray(e2)
        h = h.numpy()
        out = np.zeros(len(e2), np.float32)
        for b in np.nonzero((e2 >= self.lo) & (e2 < self.hi))[0]:
            r = e2[b] - self.lo
            out[b] = O.score_chain(h[b:b + 1], self.E[r:r + 1], self.b[r:r + 1])[0, 0]
        return torch.from_numpy(out)

    def rank_counts(self, h, tgt, e2, filt_indptr, filt_idx, filt_nnz=None, k=0):
        logits = O.score_chain(h.numpy(), self.E, self.b)
        e2, ip, ix = np.asarray(e2), np.asarray(filt_indptr), np.asarray(filt_idx)
        t = tgt.numpy()
        if t.ndim == 2:          # the [2, B] form of the product scorer: comparisons are against the exact-chain targets
            t = t[1]
        B = len(e2)
        ng, ne = np.zeros(B, np.int32), np.zeros(B, np.int32)
        for b in range(B):
            keep = np.ones(self.hi - self.lo, bool)
            f = ix[ip[b]:ip[b + 1]]
            f = f[(f >= self.lo) & (f < self.hi)] - self.lo
            keep[f] = False
            if self.lo <= e2[b] < self.hi:
                keep[e2[b] - self.lo] = False
            row = logits[b][keep]
            ng[b] = np.count_nonzero(row > t[b])
            ne[b] = np.count_nonzero(row == t[b])
        if k > 0:
            # shard-local top-k of the filtered row (target kept), global ids
            tv = np.full((B, k), -np.inf, np.float32)
            ti = np.full((B, k), -1, np.int64)
            for b in range(B):
                row = logits[b].copy()
                f = ix[ip[b]:ip[b + 1]]
                f = f[(f >= self.lo) & (f < self.hi) & (f != e2[b])] - self.lo
                row[f] = -np.inf
                order = np.lexsort((np.arange(len(row)), -row))[:k]
                order = order[row[order] > -np.inf]
                tv[b, :len(order)] = row[order]
                ti[b, :len(order)] = order + self.lo
            return torch.from_numpy(ng), torch.from_numpy(ne), torch.from_numpy(tv), torch.from_numpy(ti)
        return torch.from_numpy(ng), torch.from_numpy(ne)

    # unsharded convenience used by QueryShardedEvaluator's rank_fn
    def rank(self, h, e2, filt_indptr, filt_idx, filt_nnz=None):
        tgt = self.target_scores(h, e2)
        ng, ne = self.rank_counts(h, tgt, e2, filt_indptr, filt_idx)
        return (1 + ng).to(torch.int32), ne
