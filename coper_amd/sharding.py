"""Multi-GPU evaluation: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The reference has no distributed code at all; the path shards two ways:

* `QueryShardedEvaluator` -- small KGs (FB15k-237 / WN18RR-shaped: the whole model is a few GB and is
  replicated): the queries of a pass are split across ranks, no data-path collective; one
  all-gather of int32 ranks at the end so every rank can form the global metrics.
* `EntityShardedRanker` -- large KGs (the 10M-entity config): rank g owns entity rows
  [g|E|/G, (g+1)|E|/G) of `ent_emb` / `pred_bias`.  Per chunk of B queries, THREE collectives
  (round 2 had four):
    1. ONE all-gather of the rows each rank OWNS among `ent_emb[e1]`, `ent_emb[e2]` (with `pred_bias[e2]` beside them), padded
       to the largest share -- every row has one owner, so this moves each row once where rounds 2 - 4 all-reduced a zero-padded
       [B, 2d + 1] (a ring all-reduce moves it twice) -- plus ONE header row per rank: the largest |ent_emb| element of its rows
       and the power-of-two hint it has in force, from which every rank derives the table-wide maximum on EVERY chunk
       (a shard whose weights were reloaded can no longer run on a stale hint while the others enter collectives: ADVICE r4)
                                                                                [G, cap + 1, d + 1]  f32
    2. the encoder is split by relation (rank = rel mod G: a rank touches only its relations'
       generated dense weights); all-gather of the OWNED `h` rows, padded to the largest share
       (round 2 all-reduced a zero-padded [B, d]: G times the bytes)              [G, ceil-ish(B/G), d]  f32
       -- then every rank scores the targets itself from its copy of the e2 rows, by the fp32 chain
       (`score_rows`): the target logits are never exchanged
    3. every rank runs the fused score+count over its rows; ONE all-gather of the packed per-shard
       record (n_greater, n_equal[, top-k]) -- the collective BASELINE.json's north_star names --
       and every rank sums the counts: rank = 1 + sum_g n_greater_g (integers: identical to the
       single-GPU result).
  Payloads are <= B*(2d+1)*4 bytes (<= 8 MB), i.e. latency-bound on xGMI; logits never cross GPUs.

Both take a *scorer* returning torch tensors on its own device.  `QueryShardedEvaluator` needs `rank_pass` (or `encode` +
`rank`); `EntityShardedRanker` needs
    gather_entities(ids) -> [B, d]      gather_bias(ids) -> [B]        (0 for ids the shard does not hold)
    encode(e1, rel, e1_rows=rows) -> [B, d]
    score_rows(h, rows, bias) -> [B]    (the fp32-chain logit of rows the caller holds)
    rank_counts(h, tgt [2, B], e2, filt_indptr, filt_idx, filt_nnz=, k=) -> (n_greater, n_equal[, topk_val, topk_idx])
and, optionally, `ent_absmax()` / `set_x3_ent_absmax(v)`: the largest |ent_emb| element of the local rows and the setter of the
table-wide one -- the bf16x3 mode scales every shard's planes by the same power of two so that its logits do not depend on the
shard layout (include/coper_hip.h: x3_ent_absmax); the ranker re-derives it from the header rows of step 1 on every chunk.
Optional, for step 1 without touching the prepared state: `owned_rows(ids) -> ([n, d] rows, [n] biases)` of ids the shard holds.
`coper_amd.models.ConvE` is the product scorer; the CPU `gloo` tests (tests/test_sharding_gloo.py) plug a test-only scorer in to
exercise the exchange logic without a GPU -- the product never routes through anything else."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

__all__ = ["local_rank_pass", "QueryShardedEvaluator", "EntityShardedRanker", "shard_bounds", "merge_topk"]


def shard_bounds(num_ent: int, world: int, rank: int):
    """Rows [lo, hi) of rank `rank`: contiguous, sizes differ by at most one."""
    base, rem = divmod(int(num_ent), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def local_rank_pass(model, chunk, want_equal=False):
    """Unsharded: encode + fused filtered rank for one chunk of queries.  Returns (ranks, n_equal) int32 tensors;
    n_equal is None unless asked for (ranking_and_hits, like the reference, has no use for tie counts)."""
    if hasattr(model, "rank_pass"):    # one call per chunk (coper_encode_rank); the library itself spreads blocks with
        # thousands of known answers over the chip (k_filter_excess_bf16x3), so there is nothing to route here
        return model.rank_pass(chunk["e1"], chunk["rel"], chunk["e2"], chunk["filt_indptr"], chunk["filt_idx"],
                               filt_nnz=len(chunk["filt_idx"]), want_equal=want_equal)
    h = model.encode(chunk["e1"], chunk["rel"])   # scorers that expose only the two-call protocol
    return model.rank(h, chunk["e2"], chunk["filt_indptr"], chunk["filt_idx"], filt_nnz=len(chunk["filt_idx"]))


def _slice_chunk(chunk, lo, hi):
    ip = np.asarray(chunk["filt_indptr"])[lo:hi + 1]
    return dict(e1=chunk["e1"][lo:hi], e2=chunk["e2"][lo:hi], rel=chunk["rel"][lo:hi],
                filt_indptr=(ip - ip[0]).astype(np.int64), filt_idx=np.asarray(chunk["filt_idx"])[ip[0]:ip[-1]])


class QueryShardedEvaluator(object):
    """Replicated model, queries split contiguously across ranks."""

    def __init__(self, scorer, group=None, rank_fn=local_rank_pass):
        self.scorer, self.group, self.rank_fn = scorer, group, rank_fn
        self.dist = dist.is_initialized()      # a process group of ONE rank still runs its collectives (RCCL on one GPU)
        self.world = dist.get_world_size(group) if self.dist else 1
        self.rank_id = dist.get_rank(group) if self.dist else 0

    def rank(self, chunk):
        Q = len(chunk["e1"])
        lo, hi = shard_bounds(Q, self.world, self.rank_id)
        if hi > lo:
            r, ne = self.rank_fn(self.scorer, _slice_chunk(chunk, lo, hi), want_equal=True)
        else:
            dev = getattr(self.scorer, "device", "cpu")
            r = torch.zeros(0, dtype=torch.int32, device=dev)
            ne = torch.zeros(0, dtype=torch.int32, device=dev)
        if not self.dist:
            return r, ne
        # ragged all-gather: pad to the largest share
        cap = shard_bounds(Q, self.world, 0)[1]
        buf = torch.zeros((2, cap), dtype=torch.int32, device=r.device)
        buf[0, :hi - lo] = r
        buf[1, :hi - lo] = ne
        out = torch.empty((self.world * 2, cap), dtype=torch.int32, device=r.device)
        dist.all_gather_into_tensor(out, buf, group=self.group)
        out = out.view(self.world, 2, cap)
        rs, nes = [], []
        for g in range(self.world):
            glo, ghi = shard_bounds(Q, self.world, g)
            rs.append(out[g, 0, :ghi - glo])
            nes.append(out[g, 1, :ghi - glo])
        return torch.cat(rs), torch.cat(nes)


class EntityShardedRanker(object):
    """Entity-sharded ranking; see the module docstring for the three exchange steps."""

    def __init__(self, scorer, group=None, split_encoder=True):
        self.scorer, self.group, self.split_encoder = scorer, group, split_encoder
        self.dist = dist.is_initialized()      # a process group of ONE rank still runs its collectives (RCCL on one GPU)
        self.world = dist.get_world_size(group) if self.dist else 1
        self.rank_id = dist.get_rank(group) if self.dist else 0
        self._plan = None      # the relation split of the last chunk (encode)
        self._rows_plan = None # the ownership split of the last chunk (step 1)
        ne = getattr(scorer, "num_ent", None)
        self.bounds = np.asarray([shard_bounds(ne, self.world, g)[0] for g in range(self.world)], np.int64) if ne else None
        # one power of two for the entity planes of every shard (the x3 mode's logits are then the same bits whatever the layout):
        # agreed at construction, and again from the header rows of step 1 on every chunk (a shard whose weights were reloaded)
        self._absmax_ok = hasattr(scorer, "ent_absmax") and hasattr(scorer, "set_x3_ent_absmax")
        if self.dist and self._absmax_ok:
            m = torch.as_tensor([float(scorer.ent_absmax())], dtype=torch.float32, device=getattr(scorer, "device", "cpu"))
            dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
            scorer.set_x3_ent_absmax(float(m[0]))

    def _allreduce(self, t):
        if self.dist:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def _owned(self, ids_np):
        """(rows [n, d], biases [n]) of ids this shard holds"""
        sc = self.scorer
        if hasattr(sc, "owned_rows"):
            return sc.owned_rows(ids_np)
        return sc.gather_entities(ids_np), sc.gather_bias(ids_np)

    def _gather_rows(self, e1, e2):
        """step 1: (ent_emb[e1] [B,d], ent_emb[e2] [B,d], pred_bias[e2] [B]) from ONE all-gather of the owned rows"""
        sc = self.scorer
        # (ids are compared BY VALUE with the cached plan's: a device tensor costs a synchronising copy per chunk -- pass NumPy arrays.
        #  Identity + torch's version counter would not do: the library's own kernels write id buffers through raw pointers
        #  (coper_widen_ids, coper_stage_ids_next), which no counter sees)
        e1_np = np.asarray(e1.cpu() if isinstance(e1, torch.Tensor) else e1).astype(np.int64)
        e2_np = np.asarray(e2.cpu() if isinstance(e2, torch.Tensor) else e2).astype(np.int64)
        B = len(e1_np)
        if not self.dist or self.bounds is None:
            r1, r2, b2 = sc.gather_entities(e1_np), sc.gather_entities(e2_np), sc.gather_bias(e2_np)
            if self.dist:      # (a scorer that does not say how many entities there are: the all-reduce of rounds 2 - 4)
                d = r1.shape[1]
                pack = torch.cat([r1, r2, b2.reshape(-1, 1)], dim=1)
                self._allreduce(pack)
                return pack[:, :d].contiguous(), pack[:, d:2 * d].contiguous(), pack[:, 2 * d].contiguous()
            return r1, r2, b2
        # who owns what is known to every rank (ids are replicated): the plan is kept for the chunk it was made for
        plan = self._rows_plan
        if plan is None or plan["B"] != B or not (np.array_equal(plan["e1"], e1_np) and np.array_equal(plan["e2"], e2_np)):
            ids = np.concatenate([e1_np, e2_np])                                 # positions 0..B-1: e1, B..2B-1: e2
            owner = np.clip(np.searchsorted(self.bounds, ids, side="right") - 1, 0, self.world - 1)
            order = np.argsort(owner, kind="stable")
            counts = np.bincount(owner, minlength=self.world)
            cap = int(counts.max())
            start = np.concatenate([[0], np.cumsum(counts)])
            slot = np.empty(2 * B, np.int64)                                     # where position p sits in the gathered tensor
            for g in range(self.world):
                slot[order[start[g]:start[g + 1]]] = g * (cap + 1) + 1 + np.arange(counts[g])
            mine = order[start[self.rank_id]:start[self.rank_id + 1]]
            dev = getattr(sc, "device", "cpu")
            plan = self._rows_plan = dict(B=B, e1=e1_np.copy(), e2=e2_np.copy(), cap=cap, mine_ids=ids[mine], n_mine=len(mine),
                                          take1=torch.as_tensor(slot[:B], device=dev), take2=torch.as_tensor(slot[B:], device=dev))
        cap = plan["cap"]
        native = hasattr(sc, "pack_owned_rows") and hasattr(sc, "unpack_rows") and hasattr(sc, "shard")
        hdr = (float(sc.ent_absmax()), float(getattr(sc, "_x3_absmax", None) or 0.0)) if self._absmax_ok else (0.0, 0.0)
        if native:           # header + owned rows + zero padding: one launch (the shard-local row numbers are kept with the plan)
            if "loc" not in plan:
                plan["loc"] = torch.as_tensor(plan["mine_ids"] - int(sc.shard[0]), dtype=torch.int64).to(sc.device)
            buf = sc.pack_owned_rows(plan["loc"], cap, hdr[0], hdr[1])
            d = buf.shape[1] - 1
        else:
            rows, bias = self._owned(plan["mine_ids"])
            d = rows.shape[1]
            buf = torch.zeros((cap + 1, d + 1), dtype=torch.float32, device=rows.device)
            if self._absmax_ok:                                                  # the header row: [own maximum, hint in force]
                buf[0, 0] = hdr[0]
                buf[0, 1] = hdr[1]
            if plan["n_mine"]:
                buf[1:1 + plan["n_mine"], :d] = rows
                buf[1:1 + plan["n_mine"], d] = bias
        out = torch.empty((self.world * (cap + 1), d + 1), dtype=torch.float32, device=buf.device)
        dist.all_gather_into_tensor(out, buf, group=self.group)
        if self._absmax_ok:
            head = out.view(self.world, cap + 1, d + 1)[:, 0, :2].cpu()          # (one small D2H per chunk: every rank sees the same values)
            m = float(head[:, 0].max())
            if m > 0.0 and bool((head[:, 1] != m).any()):                        # some rank runs on another hint (reloaded weights): all re-agree
                sc.set_x3_ent_absmax(m)
        if native:
            return sc.unpack_rows(out, plan["take1"], plan["take2"])
        g1, g2 = out.index_select(0, plan["take1"]), out.index_select(0, plan["take2"])
        return g1[:, :d].contiguous(), g2[:, :d].contiguous(), g2[:, d].contiguous()

    def encode(self, e1, rel, rows=None):
        """step 2: h [B, d] on every rank; `rows` = ent_emb[e1] (step 1) or None to fetch them here"""
        sc = self.scorer
        if rows is None:
            rows = self._allreduce(sc.gather_entities(e1))
        # (host copies of the ids drive the split: pass NumPy arrays -- device tensors cost a synchronising copy each per chunk)
        rel_np = np.asarray(rel.cpu() if isinstance(rel, torch.Tensor) else rel)
        e1_np = np.asarray(e1.cpu() if isinstance(e1, torch.Tensor) else e1)
        B = len(rel_np)
        if self.world == 1 or not self.split_encoder:
            h = sc.encode(e1_np, rel_np, e1_rows=rows)
            if self.world == 1 and self.dist:          # one rank: the gather of a single share (keeps the RCCL path exercised)
                out = torch.empty_like(h)
                dist.all_gather_into_tensor(out, h.contiguous(), group=self.group)
                h = out
            return h
        # the relation split is known to every rank (ids are replicated): shares, their order, the largest one.  An evaluation
        # set is scored pass after pass: the plan (host argsort / bincount, four small H2D copies) is kept for the chunk it
        # was made for and reused while the relation ids compare equal (a 20 us check against ~200 us of planning and syncs)
        plan = self._plan
        if plan is None or plan["B"] != B or not np.array_equal(plan["rel"], rel_np):
            owner = rel_np % self.world
            order = np.argsort(owner, kind="stable")
            counts = np.bincount(owner, minlength=self.world)
            cap = int(counts.max())
            mine = order[int(counts[:self.rank_id].sum()):int(counts[:self.rank_id + 1].sum())]
            take = np.concatenate([g * cap + np.arange(counts[g]) for g in range(self.world)])
            dev = rows.device
            plan = self._plan = dict(B=B, rel=rel_np.copy(), cap=cap, mine=mine, sel=torch.as_tensor(mine, device=dev),
                                     order=torch.as_tensor(order, device=dev), take=torch.as_tensor(take, device=dev))
        cap, mine = plan["cap"], plan["mine"]
        buf = torch.zeros((cap, rows.shape[1]), dtype=torch.float32, device=rows.device)
        if len(mine):
            buf[:len(mine)] = sc.encode(e1_np[mine], rel_np[mine], e1_rows=rows.index_select(0, plan["sel"]).contiguous())
        out = torch.empty((self.world * cap, rows.shape[1]), dtype=torch.float32, device=rows.device)
        dist.all_gather_into_tensor(out, buf, group=self.group)
        # share g sits at rows [g cap, g cap + counts[g]); `order` lists the queries share by share
        h = torch.empty((B, rows.shape[1]), dtype=torch.float32, device=rows.device)
        h.index_copy_(0, plan["order"], out.index_select(0, plan["take"]))
        return h

    def rank(self, chunk, k=0):
        """Returns (ranks, n_equal) int32 [B]; with k > 0 also the global top-k of the filtered rows
        (topk_val f32 [B,k], topk_idx int64 [B,k]), merged from the per-shard top-k in (score desc, id asc) order."""
        sc = self.scorer
        rows1, rows2, bias2 = self._gather_rows(chunk["e1"], chunk["e2"])        # step 1
        h = self.encode(chunk["e1"], chunk["rel"], rows=rows1)                    # step 2
        tx = sc.score_rows(h, rows2, bias2)          # the targets by the fp32 chain, on every rank from its own copy of the rows
        tgt = torch.stack([tx, tx])
        # step 3: ONE all-gather of the packed per-shard record: [ng<<32 | ne, k score bit patterns, k ids] per query, and one more
        # row per rank: the words of its band audit (x3 scorers).  Every rank sees every rank's audit, so all of them apply the
        # SAME policy (coper_band_policy on the largest ratio): widen kappa together, and -- from 1.0 on -- count the chunk again
        # together (round 5; the unsharded ranker does the same in metrics.ranking_and_hits).  No extra collective.
        audited = hasattr(sc, "band_audit") and hasattr(sc, "band_policy") and getattr(sc, "score_mode", None) == "bf16x3"
        for attempt in range(5):
            out = sc.rank_counts(h, tgt, chunk["e2"], chunk["filt_indptr"], chunk["filt_idx"],
                                 filt_nnz=len(chunk["filt_idx"]), k=k)
            ng, ne = out[0], out[1]
            B = ng.shape[0]
            native = hasattr(sc, "pack_shard_record") and hasattr(sc, "merge_shard_records")
            if native:       # one launch: counts, top-k and the audit words (read and reset on the device: no host round trip here)
                rec = sc.pack_shard_record(ng, ne, out[2] if k > 0 else None, out[3] if k > 0 else None, reset_audit=True)
            else:
                rec = torch.zeros((B + 1, 1 + 2 * k), dtype=torch.int64, device=ng.device)
                rec[:B, 0] = (ng.to(torch.int64) << 32) | ne.to(torch.int64)
                if k > 0:
                    rec[:B, 1:1 + k] = out[2].contiguous().view(torch.int32).to(torch.int64)
                    rec[:B, 1 + k:] = out[3]
                if audited:
                    ratio, n_pairs = sc.band_audit()
                    rec[B, 0] = (int(np.float32(ratio).view(np.uint32)) << 32) | (min(int(n_pairs), 0x7fffffff) & 0xffffffff)
            if self.dist:
                allrec = torch.empty((self.world * (B + 1), 1 + 2 * k), dtype=torch.int64, device=rec.device)
                dist.all_gather_into_tensor(allrec, rec, group=self.group)           # concatenated along dim 0 (gloo + nccl)
                allrec = allrec.view(self.world, B + 1, 1 + 2 * k)
            else:
                allrec = rec.view(1, B + 1, 1 + 2 * k)
            if not audited:
                break
            words = allrec[:, B, 0].cpu().numpy()
            ratios = (words >> 32).astype(np.uint32).view(np.float32)
            pairs = int((words & 0xffffffff).sum())
            action, _ = sc.band_policy(float(ratios.max()) if pairs else 0.0, pairs)
            if action != 2:
                break
        else:
            raise RuntimeError("bf16x3 band audit: still above the band's allowance after 4 re-counted chunks")
        if native:           # one launch: the summed counts and the candidates side by side
            ranks, ne_i, vals, ids = sc.merge_shard_records(allrec, self.world, B, k)
            if k == 0:
                return ranks, ne_i
            ne_tot = ne_i
        else:
            allrec = allrec[:, :B, :]
            ng_tot = (allrec[:, :, 0] >> 32).sum(dim=0)
            ne_tot = (allrec[:, :, 0] & 0xFFFFFFFF).sum(dim=0)
            ranks = (1 + ng_tot).to(torch.int32)
            if k == 0:
                return ranks, ne_tot.to(torch.int32)
            vals = allrec[:, :, 1:1 + k].to(torch.int32).view(torch.float32).permute(1, 0, 2).reshape(B, -1)
            ids = allrec[:, :, 1 + k:].permute(1, 0, 2).reshape(B, -1)
        if self.world == 1:      # one shard: its list is already in (score desc, id asc) order
            return ranks, ne_tot.to(torch.int32), vals.contiguous(), ids.contiguous()
        tv, ti = merge_topk(vals, ids, k)
        return ranks, ne_tot.to(torch.int32), tv, ti


def merge_topk(vals, ids, k):
    """Top-k of candidate lists [B, M] in (score desc, id asc) order; padding entries are (-inf, -1)."""
    big = torch.iinfo(torch.int64).max
    key_ids = torch.where(ids < 0, torch.full_like(ids, big), ids)
    o1 = torch.argsort(key_ids, dim=1, stable=True)
    v1, i1 = torch.gather(vals, 1, o1), torch.gather(ids, 1, o1)
    o2 = torch.argsort(v1, dim=1, descending=True, stable=True)
    return torch.gather(v1, 1, o2)[:, :k].contiguous(), torch.gather(i1, 1, o2)[:, :k].contiguous()
