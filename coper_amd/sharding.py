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


class _ChunkPlan(object):
    """Host-side plan of one chunk: who owns which row (step 1), which rank encodes which query (step 2).  Every rank derives
    the same plan from the replicated ids; all index tensors travel to the device in ONE pinned buffer / one copy."""
    __slots__ = ("B", "cap1", "n_mine", "mine_ids", "cap2", "n_enc", "ints", "off", "dev", "split")


class EntityShardedRanker(object):
    """Entity-sharded ranking; see the module docstring for the three exchange steps.

    `scorer` holds the rank's entity rows and runs step 3 (targets, fused score + count, top-k); `encoder` -- by default the
    same object -- runs the rank's share of step 2.  Given a SECOND handle for the encoder (coper_config.role: one handle per
    role, no derived buffer held twice), `rank_stream` issues steps 1 - 2 of chunk n + 1 on a side stream BEFORE chunk n's
    count launch: they run under its 5.5 ms instead of in front of the next one.

    `rank(chunk)` is one chunk, checked before it returns (two small D2H reads: the header row of step 1, the audit words of
    step 3).  `rank_stream(chunks)` is the evaluation loop: plans are made one chunk ahead while the device works, nothing is
    read back inside a chunk -- the header comparison and the audit words stay on the device and are read ONCE per window of
    chunks; a window whose check fails (a shard reloaded its rows; a band too narrow) is ranked again chunk by chunk."""

    def __init__(self, scorer, group=None, split_encoder=True, encoder=None, emulate_world=None, overlap=None, side_communicator=False):
        self.scorer, self.group, self.split_encoder = scorer, group, split_encoder
        self.encoder = encoder if encoder is not None else scorer
        self.dist = dist.is_initialized()      # a process group of ONE rank still runs its collectives (RCCL on one GPU)
        self.world = dist.get_world_size(group) if self.dist else 1
        self.rank_id = dist.get_rank(group) if self.dist else 0
        # emulate_world = (G, g): ONE rank's work of a G-rank job without the other ranks (bench.py's `scale.projected`): the plans,
        # launches and buffers of rank g, every all-gather replaced by a local copy of the own share into all G slots -- the other
        # ranks' rows read as this rank's, so the VALUES mean nothing; the work per chunk is the rank's own
        self.emulated = emulate_world is not None
        if self.emulated:
            self.world, self.rank_id, self.dist = int(emulate_world[0]), int(emulate_world[1]), True
        ne = getattr(scorer, "num_ent", None)
        self.num_ent = int(ne) if ne else None
        self.bounds = np.asarray([shard_bounds(ne, self.world, g)[0] for g in range(self.world)], np.int64) if ne else None
        self.device = torch.device(getattr(scorer, "device", "cpu"))
        self.cuda = self.device.type == "cuda"
        # steps 1 - 2 of the next chunk beside this chunk's count launch: needs a handle of its own for the encoder (the two would
        # race on one handle's workspace) and a communicator of its own (one communicator runs its collectives in issue order)
        # (overlap=False: everything on one stream and one communicator, in program order -- the A/B switch, and the fallback should two
        #  communicators in flight at once ever misbehave on a node: they are issued in the same order on every rank)
        #  Default: on from two ranks on.  With ONE rank holding the whole table it costs more than it hides (measured, 10M x 256:
        #  46.8 -> 48.6 ms per chunk -- the count launch deals its rows so that the workgroups of an XCD stream one stretch of the
        #  table together, and a second kernel taking CUs in the middle of it breaks that up; on a 1.25 M-row shard 7.04 -> 6.82 ms)
        two = self.cuda and self.encoder is not self.scorer
        self.overlap = bool(two and self.dist and self.world > 1) if overlap is None else bool(overlap and two)
        self.side = torch.cuda.Stream(device=self.device) if self.overlap else None
        # The side stream's two collectives run on the SAME communicator by default: torch hands a communicator's collectives to one
        # internal stream in issue order -- [side] gather rows (n + 1), gather h (n + 1), [main] gather records (n) -- the same order on
        # every rank, and the first two have long finished when the count launch the third waits for ends.  side_communicator=True gives
        # them a communicator of their own (dist.new_group): two communicators with kernels in flight at once is the textbook way to
        # hang a node if their issue order ever differed between ranks; it does not here, but nothing in this build could run it on
        # eight GPUs, so it is opt-in.
        self.side_group = group
        if side_communicator and self.overlap and self.world > 1 and not self.emulated:
            self.side_group = dist.new_group(ranks=dist.get_process_group_ranks(group) if group is not None else None)
        self._pins, self._pin_i = [None] * 4, 0
        # one power of two for the entity planes of every shard (the x3 mode's logits are then the same bits whatever the layout):
        # agreed at construction, and again from the header rows of step 1 on every chunk (a shard whose weights were reloaded)
        self._absmax_ok = hasattr(scorer, "ent_absmax") and hasattr(scorer, "set_x3_ent_absmax")
        if self.dist and self._absmax_ok:
            m = torch.as_tensor([float(scorer.ent_absmax())], dtype=torch.float32, device=self.device)
            if not self.emulated:
                dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
            self._set_absmax(float(m[0]))

    # ------------------------------------------------------------------------------------------------ small helpers
    def _set_absmax(self, m):
        self.scorer.set_x3_ent_absmax(m)
        if self.encoder is not self.scorer and hasattr(self.encoder, "set_x3_ent_absmax"):
            self.encoder.set_x3_ent_absmax(m)

    def _allreduce(self, t):
        if self.dist and not self.emulated:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def _all_gather(self, out, buf, group):
        if self.emulated:
            out.view(self.world, -1).copy_(buf.reshape(1, -1).expand(self.world, -1))
        else:
            dist.all_gather_into_tensor(out, buf, group=group)

    def _owned(self, ids_np):
        """(rows [n, d], biases [n]) of ids this shard holds"""
        sc = self.scorer
        if hasattr(sc, "owned_rows"):
            return sc.owned_rows(ids_np)
        return sc.gather_entities(ids_np), sc.gather_bias(ids_np)

    @staticmethod
    def _host_ids(a):
        # (host copies of the ids drive the plans: pass NumPy arrays -- a device tensor costs a synchronising copy per chunk)
        return np.ascontiguousarray(np.asarray(a.cpu() if isinstance(a, torch.Tensor) else a).astype(np.int64, copy=False))

    def _hdr(self):
        sc = self.scorer
        return (float(sc.ent_absmax()), float(getattr(sc, "_x3_absmax", None) or 0.0)) if self._absmax_ok else (0.0, 0.0)

    # ------------------------------------------------------------------------------------------------ the plan (host only)
    def plan(self, chunk):
        """Everything the host contributes to a chunk, vectorised (no loop over ranks), no device call: `rank_stream` makes the
        plan of chunk n + 1 while chunk n runs.  Ids outside [0, num_ent) are refused here (ValueError) -- rounds 2 - 4 gathered
        zero rows for them, round 5 silently ranked such a query against a clamped entity; the reference's gather
        raises InvalidArgumentError."""
        e1, e2, rel = self._host_ids(chunk["e1"]), self._host_ids(chunk["e2"]), self._host_ids(chunk["rel"])
        B, G, g = len(e1), self.world, self.rank_id
        if len(e2) != B or len(rel) != B:
            raise ValueError("chunk: e1, e2, rel of one length")
        pl = _ChunkPlan()
        pl.B, pl.split = B, bool(self.split_encoder and G > 1)
        parts = []
        if self.bounds is not None and self.dist:
            ids = np.concatenate([e1, e2])                                       # positions 0..B-1: e1, B..2B-1: e2
            if B and (int(ids.min()) < 0 or int(ids.max()) >= self.num_ent):
                raise ValueError("entity id outside [0, %d) in e1 / e2" % self.num_ent)
            owner = np.searchsorted(self.bounds, ids, side="right") - 1
            order = np.argsort(owner, kind="stable")
            counts = np.bincount(owner, minlength=G)
            start = np.concatenate([[0], np.cumsum(counts)])
            cap = int(counts.max()) if B else 0
            so = owner[order]
            slot = np.empty(2 * B, np.int64)                                     # where position p sits in the gathered tensor
            slot[order] = so * (cap + 1) + 1 + (np.arange(2 * B) - start[so])
            mine = order[start[g]:start[g + 1]]
            pl.cap1, pl.n_mine, pl.mine_ids = cap, len(mine), ids[mine]
            parts += [("take1", slot[:B]), ("take2", slot[B:]), ("loc", pl.mine_ids - int(self.bounds[g]))]
        else:
            pl.cap1, pl.n_mine, pl.mine_ids = 0, 0, None
        if pl.split:
            owner = rel % G
            order = np.argsort(owner, kind="stable")
            counts = np.bincount(owner, minlength=G)
            start = np.concatenate([[0], np.cumsum(counts)])
            cap = int(counts.max()) if B else 0
            so = owner[order]
            mine = order[start[g]:start[g + 1]]
            pl.cap2, pl.n_enc = cap, len(mine)
            # share g sits at rows [g cap, g cap + counts[g]) of the gathered h; `order` lists the queries share by share
            parts += [("sel", mine), ("order", order), ("take", so * cap + (np.arange(B) - start[so])), ("rel_mine", rel[mine])]
        else:
            pl.cap2, pl.n_enc = B, B
            parts += [("rel_all", rel)]
        pl.off, n = {}, 0
        for name, a in parts:
            pl.off[name] = (n, n + len(a))
            n += len(a)
        if self.cuda:       # one pinned buffer (a small ring: the copy of the plan before last may still be in flight), one H2D
            i = self._pin_i = (self._pin_i + 1) % len(self._pins)
            if self._pins[i] is None or self._pins[i].numel() < max(1, n):
                self._pins[i] = torch.empty(max(1, n) * 2, dtype=torch.int64).pin_memory()
            pl.ints = self._pins[i][:n]
            host = pl.ints.numpy()
        else:
            pl.ints = torch.empty(n, dtype=torch.int64)
            host = pl.ints.numpy()
        for name, a in parts:
            lo, hi = pl.off[name]
            host[lo:hi] = a
        pl.dev = None
        return pl

    def _dev(self, pl, name):
        if pl.dev is None:
            pl.dev = pl.ints.to(self.device, non_blocking=True) if self.cuda else pl.ints
        lo, hi = pl.off[name]
        return pl.dev[lo:hi]

    # ------------------------------------------------------------------------------------------------ steps 1 and 2
    def _steps12(self, pl, chunk, group):
        """(h [B, d], ent_emb[e2] [B, d], pred_bias[e2] [B], header state [2] or None) on the current stream.  The header state --
        {table-wide maximum seen in the header rows, 1.0 when some rank runs on another hint} -- stays on the device."""
        sc, enc, G, B = self.scorer, self.encoder, self.world, pl.B
        hdr_state = None
        if pl.mine_ids is None:      # not distributed, or a scorer that does not say how many entities there are (the all-reduce of rounds 2 - 4)
            e1, e2 = self._host_ids(chunk["e1"]), self._host_ids(chunk["e2"])
            rows1, rows2, bias2 = sc.gather_entities(e1), sc.gather_entities(e2), sc.gather_bias(e2)
            if self.dist:
                d = rows1.shape[1]
                pack = torch.cat([rows1, rows2, bias2.reshape(-1, 1)], dim=1)
                self._allreduce(pack)
                rows1, rows2, bias2 = pack[:, :d].contiguous(), pack[:, d:2 * d].contiguous(), pack[:, 2 * d].contiguous()
        else:
            cap = pl.cap1
            native = hasattr(sc, "pack_owned_rows") and hasattr(sc, "unpack_rows") and hasattr(sc, "shard")
            hdr = self._hdr()
            if native:           # header + owned rows + zero padding: one launch
                buf = sc.pack_owned_rows(self._dev(pl, "loc"), cap, hdr[0], hdr[1])
                d = buf.shape[1] - 1
            else:
                rows, bias = self._owned(pl.mine_ids)
                d = rows.shape[1]
                buf = torch.zeros((cap + 1, d + 1), dtype=torch.float32, device=rows.device)
                buf[0, 0], buf[0, 1] = hdr[0], hdr[1]                            # the header row: [own maximum, hint in force]
                if pl.n_mine:
                    buf[1:1 + pl.n_mine, :d] = rows
                    buf[1:1 + pl.n_mine, d] = bias
            out = torch.empty((G * (cap + 1), d + 1), dtype=torch.float32, device=buf.device)
            self._all_gather(out, buf, group)
            if self._absmax_ok:
                head = out.view(G, cap + 1, d + 1)[:, 0, :2]
                m = head[:, 0].max()
                hdr_state = torch.stack([m, ((head[:, 1] != m).any() & (m > 0)).to(torch.float32)])
            if native:
                rows1, rows2, bias2 = sc.unpack_rows(out, self._dev(pl, "take1"), self._dev(pl, "take2"))
            else:
                g1, g2 = out.index_select(0, self._dev(pl, "take1")), out.index_select(0, self._dev(pl, "take2"))
                rows1, rows2, bias2 = g1[:, :d].contiguous(), g2[:, :d].contiguous(), g2[:, d].contiguous()
        # step 2
        if not pl.split:
            h = enc.encode(None, self._dev(pl, "rel_all"), e1_rows=rows1)
            if G == 1 and self.dist:          # one rank: the gather of a single share (keeps the RCCL path exercised)
                out = torch.empty_like(h)
                self._all_gather(out, h.contiguous(), group)
                h = out
        else:
            d = rows1.shape[1]
            buf = torch.zeros((pl.cap2, d), dtype=torch.float32, device=rows1.device)
            if pl.n_enc:
                buf[:pl.n_enc] = enc.encode(None, self._dev(pl, "rel_mine"), e1_rows=rows1.index_select(0, self._dev(pl, "sel")).contiguous())
            out = torch.empty((G * pl.cap2, d), dtype=torch.float32, device=rows1.device)
            self._all_gather(out, buf, group)
            h = torch.empty((B, d), dtype=torch.float32, device=rows1.device)
            h.index_copy_(0, self._dev(pl, "order"), out.index_select(0, self._dev(pl, "take")))
        return h, rows2, bias2, hdr_state

    # ------------------------------------------------------------------------------------------------ step 3
    def _audited(self):
        sc = self.scorer
        return hasattr(sc, "band_audit") and hasattr(sc, "band_policy") and getattr(sc, "score_mode", None) == "bf16x3"

    def _step3(self, chunk, h, rows2, bias2, k):
        """The gathered records [G, B + 1, 1 + 2 k] (device) of one count over the rank's rows: ONE all-gather of the packed
        per-shard record [ng << 32 | ne, k score bit patterns, k ids] per query, and one more row per rank: the words of its band
        audit (x3 scorers) -- every rank sees every rank's audit, so all of them apply the SAME policy without another collective."""
        sc = self.scorer
        tx = sc.score_rows(h, rows2, bias2)          # the targets by the fp32 chain, on every rank from its own copy of the rows
        tgt = torch.stack([tx, tx])
        # (`e2_dev`: the targets' ids already on the device -- the plans read chunk["e2"] on the host)
        out = sc.rank_counts(h, tgt, chunk.get("e2_dev", chunk["e2"]), chunk["filt_indptr"], chunk["filt_idx"], filt_nnz=len(chunk["filt_idx"]), k=k)
        ng, ne = out[0], out[1]
        B = ng.shape[0]
        if hasattr(sc, "pack_shard_record"):     # one launch: counts, top-k and the audit words (read and reset on the device)
            rec = sc.pack_shard_record(ng, ne, out[2] if k > 0 else None, out[3] if k > 0 else None, reset_audit=True)
        else:
            rec = torch.zeros((B + 1, 1 + 2 * k), dtype=torch.int64, device=ng.device)
            rec[:B, 0] = (ng.to(torch.int64) << 32) | ne.to(torch.int64)
            if k > 0:
                rec[:B, 1:1 + k] = out[2].contiguous().view(torch.int32).to(torch.int64)
                rec[:B, 1 + k:] = out[3]
            if self._audited():
                ratio, n_pairs = sc.band_audit()
                rec[B, 0] = (int(np.float32(ratio).view(np.uint32)) << 32) | (min(int(n_pairs), 0x7fffffff) & 0xffffffff)
        if self.dist:
            allrec = torch.empty((self.world * (B + 1), 1 + 2 * k), dtype=torch.int64, device=rec.device)
            self._all_gather(allrec, rec, self.group)                            # concatenated along dim 0 (gloo + nccl)
            return allrec.view(self.world, B + 1, 1 + 2 * k)
        return rec.view(1, B + 1, 1 + 2 * k)

    def _merge(self, allrec, B, k):
        sc = self.scorer
        if hasattr(sc, "merge_shard_records"):   # one launch: the summed counts and the candidates side by side
            ranks, ne_tot, vals, ids = sc.merge_shard_records(allrec, self.world, B, k)
            if k == 0:
                return ranks, ne_tot
        else:
            rec = allrec[:, :B, :]
            ranks = (1 + (rec[:, :, 0] >> 32).sum(dim=0)).to(torch.int32)
            ne_tot = (rec[:, :, 0] & 0xFFFFFFFF).sum(dim=0).to(torch.int32)
            if k == 0:
                return ranks, ne_tot
            vals = rec[:, :, 1:1 + k].to(torch.int32).view(torch.float32).permute(1, 0, 2).reshape(B, -1)
            ids = rec[:, :, 1 + k:].permute(1, 0, 2).reshape(B, -1)
        if self.world == 1:      # one shard: its list is already in (score desc, id asc) order
            return ranks, ne_tot.to(torch.int32), vals.contiguous(), ids.contiguous()
        tv, ti = merge_topk(vals, ids, k)
        return ranks, ne_tot.to(torch.int32), tv, ti

    @staticmethod
    def _audit_words(words):
        """(largest ratio, pairs in all, per-rank ratios) of the audit words [.., G] (host int64 array)"""
        ratios = (words >> 32).astype(np.uint32).view(np.float32)
        return ratios, (words & 0xffffffff)

    # ------------------------------------------------------------------------------------------------ one chunk, checked
    def rank(self, chunk, k=0):
        """Returns (ranks, n_equal) int32 [B]; with k > 0 also the global top-k of the filtered rows
        (topk_val f32 [B,k], topk_idx int64 [B,k]), merged from the per-shard top-k in (score desc, id asc) order."""
        pl = self.plan(chunk)
        B = pl.B
        h, rows2, bias2, hdr_state = self._steps12(pl, chunk, self.group)
        if hdr_state is not None:
            st = hdr_state.cpu()                     # (one small D2H per chunk: every rank sees the same values)
            if float(st[1]) != 0.0:                  # some rank runs on another hint (reloaded weights): all re-agree, h is encoded again
                self._set_absmax(float(st[0]))
                h, rows2, bias2, _ = self._steps12(pl, chunk, self.group)
        audited = self._audited()
        for attempt in range(5):
            allrec = self._step3(chunk, h, rows2, bias2, k)
            if not audited:
                break
            ratios, pairs = self._audit_words(allrec[:, B, 0].cpu().numpy())
            n_pairs = int(pairs.sum())
            action, _ = self.scorer.band_policy(float(ratios.max()) if n_pairs else 0.0, n_pairs)
            if action != 2:
                break
        else:
            raise RuntimeError("bf16x3 band audit: still above the band's allowance after 4 re-counted chunks")
        return self._merge(allrec, B, k)

    # ------------------------------------------------------------------------------------------------ the evaluation loop
    def rank_stream(self, chunks, k=0, window=8):
        """Generator over `chunks` (an iterable of chunk dicts: e1 / rel / e2 host arrays, the CSR filter on the host or the
        device): yields what `rank` returns for each, in order.  Per chunk the host only plans (one chunk ahead) and enqueues;
        header comparisons and audit words are read once per `window` chunks, and the results of a window are yielded after its
        check -- a window that fails it is ranked again chunk by chunk with `rank` (every rank takes the same decision: the words
        are the same on all of them)."""
        it = iter(chunks)
        main = torch.cuda.current_stream(self.device) if self.cuda else None
        audited = self._audited()

        def issue12(chunk):
            pl = self.plan(chunk)                                  # host work: overlaps whatever the device is doing
            if self.overlap:
                self.side.wait_stream(main)                        # (the parameters / the last set_absmax are stream-ordered on main)
                with torch.cuda.stream(self.side):
                    res = self._steps12(pl, chunk, self.side_group)
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                for t in res[:3]:
                    t.record_stream(main)
                if res[3] is not None:
                    res[3].record_stream(main)
                return chunk, pl, res, ev
            return chunk, pl, self._steps12(pl, chunk, self.group), None

        nxt_chunk = next(it, None)
        nxt = issue12(nxt_chunk) if nxt_chunk is not None else None
        pending = []
        while nxt is not None:
            chunk, pl, (h, rows2, bias2, hdr_state), ev = nxt
            # steps 1 - 2 of the FOLLOWING chunk first: with a side stream they run under this chunk's count launch
            nxt_chunk = next(it, None)
            nxt = issue12(nxt_chunk) if nxt_chunk is not None else None
            if ev is not None:
                main.wait_event(ev)
            allrec = self._step3(chunk, h, rows2, bias2, k)
            pending.append((chunk, pl.B, allrec, hdr_state, self._merge(allrec, pl.B, k)))
            if len(pending) >= window or nxt is None:
                redo_all, redo = False, set()
                states = [p[3] for p in pending if p[3] is not None]
                words = [p[2][:, p[1], 0] for p in pending] if audited else []
                if states or words:                                 # ONE wait per window (two small copies behind the same work)
                    st = torch.stack(states).cpu().numpy() if states else None
                    wd = torch.stack(words).cpu().numpy() if words else None
                    if st is not None and (st[:, 1] != 0).any():
                        self._set_absmax(float(st[:, 0].max()))
                        redo_all = True
                    if wd is not None and not redo_all:
                        ratios, pairs = self._audit_words(wd)       # [n, G]
                        n_pairs = int(pairs.sum())
                        action, _ = self.scorer.band_policy(float(ratios.max()) if n_pairs else 0.0, n_pairs)
                        if action == 2:
                            redo = {i for i in range(len(pending)) if float(ratios[i].max()) >= 1.0}
                if (redo_all or redo) and nxt is not None and self.overlap:
                    main.wait_stream(self.side)                     # (the chunk issued ahead ran on the old planes / hint: issued again below)
                for i, p in enumerate(pending):
                    yield self.rank(p[0], k=k) if (redo_all or i in redo) else p[4]
                if redo_all and nxt is not None:
                    nxt = issue12(nxt[0])
                pending = []


def merge_topk(vals, ids, k):
    """Top-k of candidate lists [B, M] in (score desc, id asc) order; padding entries are (-inf, -1)."""
    big = torch.iinfo(torch.int64).max
    key_ids = torch.where(ids < 0, torch.full_like(ids, big), ids)
    o1 = torch.argsort(key_ids, dim=1, stable=True)
    v1, i1 = torch.gather(vals, 1, o1), torch.gather(ids, 1, o1)
    o2 = torch.argsort(v1, dim=1, descending=True, stable=True)
    return torch.gather(v1, 1, o2)[:, :k].contiguous(), torch.gather(i1, 1, o2)[:, :k].contiguous()
