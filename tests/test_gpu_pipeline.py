"""GPU: the pass pipeline as a state machine (VERDICT r5 "Next round" 2).

Three kinds of jobs ride in later launches of a handle (coper_stage_ids_next, coper_post_i32_next, coper_group_next) and a
grouping prepared ahead is consumed by pointer identity.  Two things are tested here:

  * THE GUARD: ids rewritten between the launch that sorted a batch and the pass that consumes the sorting are noticed on the
    device -- every rank of that pass comes back as COPER_RANK_STALE, the pass is counted, `RankStream` ranks the batch again
    and returns the ranks of the ids that are THERE (round 5 returned the ranks of the ids that were there: wrong, no error);
  * RANDOM INTERLEAVINGS of stage_next / group_next / post_next / encode / rank_pass (both rank paths) / prepare / train_step /
    graph replays / a growing workspace, with the staged ids rewritten at random points: every pass's ranks are those of an
    un-pipelined handle on the ids that are in the array at that moment -- or all COPER_RANK_STALE, and that only when the array
    really changed after its sorting was launched (no false alarms, no wrong ranks)."""
import os

import numpy as np
import pytest
import torch

from coper_amd import _lib
from coper_amd import data as cdata

pytestmark = pytest.mark.gpu
KEYS = ("e1", "rel", "e2", "filt_indptr", "filt_idx")


def _model(md, p, **kw):
    from coper_amd.models import ConvE
    return ConvE(md, device="cuda:0", score_mode="bf16x3", **kw).load_parameters(p).prepare()


def _plain(m, q):
    return m.rank_pass(q["e1"], q["rel"], q["e2"], q["filt_indptr"], q["filt_idx"], want_equal=False)[0].cpu().numpy()


@pytest.mark.parametrize("workload,Q", [("fb15k237_cpg", 5000), ("fb15k237_cpg", 600), ("fb15k237_plain", 2500), ("wn18rr_cpg", 700)])
def test_a_stale_prepared_grouping_never_reaches_the_ranks(workload, Q):
    from coper_amd.stream import RankStream
    md = cdata.model_descriptors(workload, num_ent=3000)
    p = cdata.synthetic_params(md, 3)
    m = _model(md, p)
    qs = [cdata.synthetic_queries(md, Q, seed=70 + i) for i in range(4)]
    base = [_plain(m, q) for q in qs]
    nnz_cap = max(len(q["filt_idx"]) for q in qs)
    rs = RankStream(m, Q, nnz_cap)
    assert m.stale_passes() == 0
    # an undisturbed stream: the ranks of plain passes, nothing stale, ONE grouping launch (the first pass's)
    m.profile(True); m.profile_read("group")
    got = rs.run(qs)
    for g, b in zip(got, base):
        assert np.array_equal(g, b)
    assert rs.stale_passes == 0 and m.stale_passes() == 0 and m.profile_read("group")[1] == 1
    # batch 2 is staged and sorted beside pass 1; then its e1 / rel are REWRITTEN in the staging array (batch 3's ids: a host that
    # refills the buffer too early) before pass 2 runs.  Round 5: pass 2 encoded the ids that WERE there and nobody knew.  Now the
    # pass is found stale on the device, counted, and ranked again from the host's copy: the ranks of batch 2.
    pk = [rs.pack(q) for q in qs]

    def disturb(n):
        if n == 1:                                            # (pass 1 is queued; its launch brings batch 2 into stages[0])
            v = rs._views(0, pk[2])
            v[0].copy_(torch.as_tensor(qs[3]["e1"]).to("cuda:0")); v[1].copy_(torch.as_tensor(qs[3]["rel"]).to("cuda:0"))
    got = rs.run(pk[:3], on_pass=disturb)
    assert rs.stale_passes == 1 and m.stale_passes() == 0     # (counted by the stream, the library's counter read and reset)
    for n in range(3):
        assert np.array_equal(got[n], base[n]), n
    # the same disturbance without the grouping done ahead: every pass sorts the ids that are there -- pass 2 ranks batch 3's queries
    # against batch 2's targets, which is what the caller wrote; nothing is stale
    mixed = dict(qs[2]); mixed["e1"], mixed["rel"] = qs[3]["e1"], qs[3]["rel"]
    want = _plain(m, mixed)
    got = rs.run(pk[:3], group_ahead=False, on_pass=disturb)
    assert rs.stale_passes == 1 and np.array_equal(got[2], want)
    m.close()


@pytest.mark.parametrize("workload,Q", [("fb15k237_cpg", 5000), ("fb15k237_cpg", 600), ("fb15k237_plain", 2500), ("wn18rr_cpg", 700)])
def test_the_guard_through_the_c_abi(workload, Q):
    """coper_group_next / coper_encode_rank / coper_stale_passes directly: one changed relation id, one changed entity id, a
    changed e2 (not a grouping input: nothing stale), the n_equal path, e1_rows, and an id changed back before the pass."""
    md = cdata.model_descriptors(workload, num_ent=3000)
    p = cdata.synthetic_params(md, 4)
    m = _model(md, p)
    qa, qb = cdata.synthetic_queries(md, Q, seed=11), cdata.synthetic_queries(md, Q, seed=12)
    da = {k: torch.as_tensor(np.asarray(v)).to("cuda:0") for k, v in qa.items()}
    db = {k: torch.as_tensor(np.asarray(v)).to("cuda:0") for k, v in qb.items()}
    base_b = _plain(m, qb)

    def run_a_then_b(want_equal=False, rows=None):
        if rows is None:
            m.group_next(db["e1"], db["rel"])
        else:
            m.group_next(None, db["rel"], e1_rows=True)
        m.rank_pass(da["e1"], da["rel"], da["e2"], da["filt_indptr"], da["filt_idx"], want_equal=False)

    def pass_b(want_equal=False, rows=None):
        return m.rank_pass(None if rows is not None else db["e1"], db["rel"], db["e2"], db["filt_indptr"], db["filt_idx"],
                           want_equal=want_equal, e1_rows=rows)[0].cpu().numpy()

    m.profile(True)
    for want_equal in (False, True):
        # undisturbed: consumed (no grouping launch), right ranks, nothing counted
        run_a_then_b()
        m.profile_read("group")
        assert np.array_equal(pass_b(want_equal), base_b) and m.profile_read("group")[1] == 0 and m.stale_passes() == 0
        # one relation id changed (to another valid id): stale, every rank negative, counted once; the plain pass that follows is right
        for key, pos in (("rel", Q // 3), ("e1", Q - 1)):
            run_a_then_b()
            old = int(db[key][pos])
            new = (old + 1) % (md["num_rel"] if key == "rel" else md["num_ent"])
            qv = {k: np.array(v, copy=True) for k, v in qb.items()}
            qv[key][pos] = new
            want = _plain(m, qv)
            run_a_then_b()                                     # (the plain pass above dropped the prepared grouping: prepare it again)
            db[key][pos] = new
            got = pass_b(want_equal)
            assert (got < 0).all() and got.max() <= _lib.RANK_STALE + 10 ** 8, (key, got[:4])
            assert m.stale_passes() == 1 and m.stale_passes() == 0
            assert np.array_equal(pass_b(want_equal), want)    # the pass after a stale one groups itself
            db[key][pos] = old
        # e2 / the filter are not grouping inputs: a changed target is simply ranked
        run_a_then_b()
        qv = {k: np.array(v, copy=True) for k, v in qb.items()}
        qv["e2"][5] = (qv["e2"][5] + 7) % md["num_ent"]
        want = _plain(m, qv)
        run_a_then_b()
        old = int(db["e2"][5]); db["e2"][5] = int(qv["e2"][5])
        assert np.array_equal(pass_b(want_equal), want) and m.stale_passes() == 0
        db["e2"][5] = old
        # changed and changed back before the pass: the ids that are there are the ids that were sorted
        run_a_then_b()
        old = int(db["rel"][0]); db["rel"][0] = (old + 1) % md["num_rel"]; db["rel"][0] = old
        assert np.array_equal(pass_b(want_equal), base_b) and m.stale_passes() == 0
    # e1_rows: only rel is a grouping input
    rows_a, rows_b = m.gather_entities(da["e1"]), m.gather_entities(db["e1"])
    m.group_next(None, db["rel"], e1_rows=True)
    m.rank_pass(None, da["rel"], da["e2"], da["filt_indptr"], da["filt_idx"], want_equal=False, e1_rows=rows_a)
    assert np.array_equal(pass_b(False, rows_b), base_b) and m.stale_passes() == 0
    m.group_next(None, db["rel"], e1_rows=True)
    m.rank_pass(None, da["rel"], da["e2"], da["filt_indptr"], da["filt_idx"], want_equal=False, e1_rows=rows_a)
    old = int(db["rel"][7]); db["rel"][7] = (old + 1) % md["num_rel"]
    got = pass_b(False, rows_b)
    assert (got < 0).all() and m.stale_passes() == 1
    db["rel"][7] = old
    # coper_encode never runs on a prepared grouping (it writes no ranks that could carry the verdict): it groups itself
    run_a_then_b()
    m.profile_read("group")
    h1 = m.encode(db["e1"], db["rel"])
    assert m.profile_read("group")[1] == 1
    h0 = m.encode(qb["e1"], qb["rel"])
    assert torch.equal(h0, h1)
    m.close()


# ------------------------------------------------------------------------------------------------------------------------------
# random interleavings
# ------------------------------------------------------------------------------------------------------------------------------
_TRAIN_MD = dict(num_ent=700, num_rel=6, ent_emb_size=200, rel_emb_size=8, emb_h=10, emb_w=20, conv_num_channels=32,
                 context_rel_conv=None, context_rel_out=[])


def _train_batch(md, B, L, seed):
    rng = np.random.default_rng(seed)
    E, R = md["num_ent"], md["num_rel"]
    lookup = rng.integers(0, E, (B, L)).astype(np.int32)
    labels = np.zeros((B, L), np.float32)
    labels[:, 0] = 1.0
    return dict(e1=rng.integers(0, E, B), rel=rng.integers(0, R, B), lookup_values=lookup, e2_multi=labels)


class _Machine(object):
    """The handle under test, a second handle that never pipelines anything (the checker: same parameter tensors), and the little
    the test has to know about the pipeline to say what every pass must return: which batch an array holds, which jobs are
    pending, and whether an array changed after the launch that sorted it."""

    def __init__(self, md, p, Q, n_batches, seed, train):
        from coper_amd.models import ConvE
        self.md, self.Q, self.rng = md, Q, np.random.default_rng(seed)
        tens = {k: torch.as_tensor(np.array(v, np.float32)).to("cuda:0") for k, v in p.items()}
        self.m = ConvE(md, device="cuda:0", score_mode="bf16x3").load_parameters(tens)
        self.ref = ConvE(md, device="cuda:0", score_mode="bf16x3").load_parameters(tens)
        self.train = train
        if train:
            self.m.train_init(seed=5)
        self.m.prepare(); self.ref.prepare()
        qs = [cdata.synthetic_queries(md, Q, seed=900 + 17 * seed + i) for i in range(n_batches)]
        # two one-id variants: the smallest rewrite a host can make
        v1 = {k: np.array(v, copy=True) for k, v in qs[0].items()}
        v1["rel"][Q // 2] = (v1["rel"][Q // 2] + 1) % md["num_rel"]
        v2 = {k: np.array(v, copy=True) for k, v in qs[1].items()}
        v2["e1"][0] = (v2["e1"][0] + 1) % md["num_ent"]
        self.qs = qs + [v1, v2]
        self.cap = max(len(q["filt_idx"]) for q in self.qs)
        sizes = [Q, Q, Q, Q + 1, self.cap]
        self.offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self.pins = []
        for q in self.qs:
            pin = torch.zeros(int(self.offs[-1]), dtype=torch.int32).pin_memory()
            for k, o in zip(KEYS, self.offs):
                a = np.asarray(q[k]).astype(np.int32)
                pin[o:o + len(a)] = torch.as_tensor(a)
            self.pins.append(pin)
        self.stages = [torch.zeros(int(self.offs[-1]), dtype=torch.int64, device="cuda:0") for _ in range(2)]
        self.views = [{k: st[o:o + n] for k, o, n in zip(KEYS, self.offs, sizes)} for st in self.stages]
        self.m.reserve(Q, self.cap)
        self.ws_q = Q
        self.content = [None, None]            # which batch an array holds
        self.changed_at = [0, 0]               # the clock of its last change
        self.sort_clock = [None, None]         # the clock of the launch that carried (or would have carried) a sorting of it
        self.clock = 0                         # eager passes so far
        self.pending_stage = None              # (array, batch)
        self.pending_group = None              # array
        self.pending_post = None               # (device ranks, host buffer, expected ndarray or None = stale allowed)
        self.posted = []
        self._base = {}
        self._base_h = {}
        self.graph = None
        self.n_stale = 0
        for c in (0, 1):
            self.write(c, int(self.rng.integers(len(self.qs))))

    def close(self):
        self.m.close(); self.ref.close()

    # ---- the checker
    def base(self, b):
        if b not in self._base:
            self._base[b] = _plain(self.ref, self.qs[b])
        return self._base[b]

    def base_h(self, b):
        if b not in self._base_h:
            self._base_h[b] = self.ref.encode(self.qs[b]["e1"], self.qs[b]["rel"]).clone()
        return self._base_h[b]

    # ---- operations
    def write(self, c, b):
        """an eager rewrite of array c (coper_widen_ids: stream-ordered, like any host that refills a staging buffer)"""
        self.m.widen_ids(self.pins[b], out=self.stages[c])
        if self.content[c] != b:
            self.changed_at[c] = self.clock + 0.5          # after the passes so far, before the next one
        self.content[c] = b

    def stage_next(self, c, b):
        self.m.stage_next(self.pins[b], self.stages[c])
        self.pending_stage = (c, b)

    def group_next(self, c):
        v = self.views[c]
        self.m.group_next(v["e1"], v["rel"])
        self.pending_group = c

    def _eager_call_begins(self):
        """what any eager encode / rank call carries; returns the sorting clocks this call may consume (a grouping prepared ahead
        is for the call right after the launch that sorted it, whatever that call is)"""
        self.clock += 1
        if self.pending_stage is not None:
            c, b = self.pending_stage
            if self.content[c] != b:
                self.changed_at[c] = self.clock            # written inside this launch (the sorting of the same launch waits for it)
            self.content[c] = b
            self.pending_stage = None
        consumable, self.sort_clock = self.sort_clock, [None, None]
        if self.pending_group is not None:
            self.sort_clock[self.pending_group] = self.clock
            self.pending_group = None
        if self.pending_post is not None:
            self.posted.append(self.pending_post)
            self.pending_post = None
        return consumable

    def rank(self, c, want_equal):
        if self.pending_stage is not None and self.pending_stage[0] == c:
            return                                        # (a pass must not read the array its own launch refills)
        consumable = self._eager_call_begins()
        v, b = self.views[c], self.content[c]
        nnz = len(self.qs[b]["filt_idx"])
        out = torch.empty(self.Q, dtype=torch.int32, device="cuda:0")
        r, _ = self.m.rank_pass(v["e1"], v["rel"], v["e2"], v["filt_indptr"], v["filt_idx"][:nnz], filt_nnz=nnz, want_equal=want_equal, out=out)
        may_be_stale = consumable[c] is not None and self.changed_at[c] > consumable[c]
        if self.rng.random() < 0.5:
            host = torch.full((self.Q,), -7, dtype=torch.int32).pin_memory()
            self.m.post_next(r, host)
            self.pending_post = (host, self.base(b), may_be_stale, (c, b))      # (what the checker says NOW: a training step may follow)
        else:
            self.check(r.cpu().numpy(), self.base(b), may_be_stale, (c, b))

    def check(self, got, want, may_be_stale, where):
        if (got < 1).any():
            assert (got < 0).all(), "a mixture of ranks and COPER_RANK_STALE"
            assert may_be_stale, "a pass reported a stale grouping although array %d (batch %d) had not changed since it was sorted" % where
            self.n_stale += 1
        else:
            assert np.array_equal(got, want), "ranks differ from the un-pipelined handle's (array %d, batch %d)" % where

    def encode(self, c):
        if self.pending_stage is not None and self.pending_stage[0] == c:
            return
        self._eager_call_begins()
        v, b = self.views[c], self.content[c]
        h = self.m.encode(v["e1"], v["rel"])
        assert torch.equal(h, self.base_h(b))

    def prepare(self):
        self.m._prepared = False
        self.m.prepare()
        self.sort_clock = [None, None]
        self.pending_group = None
        self.graph = None                                 # (a captured pass holds the derived buffers of its capture)

    def train_step(self):
        self.m.train_step(_train_batch(self.md, 32, 20, int(self.rng.integers(1 << 30))))
        torch.cuda.synchronize()
        self.ref._prepared = False
        self.ref.prepare()                                # (the checker follows the weights)
        self._base.clear(); self._base_h.clear()
        self.sort_clock = [None, None]
        self.pending_group = None
        self.graph = None                                 # (a captured pass holds the generated weights of its capture)

    def replay(self):
        if self.graph is None:
            self.graph = self.m.capture_rank_pass(self.Q, self.cap, want_equal=False)
            self.ws_q = max(self.ws_q, self.Q)
            # capturing ran eager warm-up passes: they carried whatever was pending
            self._eager_call_begins()
            self.sort_clock = [None, None]
        b = int(self.rng.integers(len(self.qs)))
        q = self.qs[b]
        got = self.graph(q["e1"], q["rel"], q["e2"], q["filt_indptr"], q["filt_idx"])[0].cpu().numpy()
        assert np.array_equal(got, self.base(b))
        # (a replay consumes nothing and carries nothing; a grouping prepared ahead stays for the next eager call)

    def grow(self):
        self.ws_q += 96
        self.m.reserve(self.ws_q, self.cap)
        self.sort_clock = [None, None]
        self.pending_group = None
        self.graph = None                                 # (captured for the old workspace)

    def finish(self):
        self.m.post_flush()
        torch.cuda.synchronize()
        if self.pending_post is not None:
            self.posted.append(self.pending_post)
            self.pending_post = None
        for host, want, may_be_stale, where in self.posted:
            self.check(host.numpy(), want, may_be_stale, where)
        self.posted = []
        counted = self.m.stale_passes()
        assert counted == self.n_stale, (counted, self.n_stale)
        self.n_stale = 0
        return counted


def _run_sequences(md, p, Q, n_seq, n_ops, seed, train=False):
    mc = _Machine(md, p, Q, 4, seed, train)
    rng = mc.rng
    ops = ["rank", "rank", "rank", "rank_ne", "encode", "stage_next", "group_next", "pipeline", "write", "prepare", "replay", "grow", "flush", "behind"]
    weights = np.array([6, 6, 6, 3, 2, 5, 6, 8, 4, 1, 1.5, 0.3, 2, 1.5])
    if train:
        ops.append("train"); weights = np.append(weights, 1.0)
    weights = weights / weights.sum()
    n_stale_seen = 0
    for s in range(n_seq):
        trace = []
        try:
            for _ in range(n_ops):
                op = str(rng.choice(ops, p=weights))
                c, b = int(rng.integers(2)), int(rng.integers(len(mc.qs)))
                trace.append((op, c, b))
                if op == "rank":
                    mc.rank(c, False)
                elif op == "rank_ne":
                    mc.rank(c, True)
                elif op == "encode":
                    mc.encode(c)
                elif op == "stage_next":
                    mc.stage_next(c, b)
                elif op == "group_next":
                    mc.group_next(c)
                elif op == "pipeline":                     # the intended use: next batch staged + sorted beside this pass
                    mc.stage_next(1 - c, b); mc.group_next(1 - c); mc.rank(c, False); mc.rank(1 - c, False)
                elif op == "behind":                       # the mistake the guard is for: the array refilled behind its sorting
                    mc.group_next(c); mc.rank(1 - c, False); mc.write(c, b); mc.rank(c, bool(b & 1))
                elif op == "write":
                    mc.write(c, b)
                elif op == "prepare":
                    mc.prepare()
                elif op == "replay":
                    mc.replay()
                elif op == "grow":
                    mc.grow()
                elif op == "train":
                    mc.train_step()
                elif op == "flush":
                    n_stale_seen += mc.finish()
            n_stale_seen += mc.finish()
        except AssertionError as e:
            raise AssertionError("sequence %d (seed %d): %s\n%s" % (s, seed, e, trace[-25:]))
    mc.close()
    return n_stale_seen


@pytest.mark.parametrize("workload,Q,seed", [("fb15k237_cpg", 4500, 1), ("fb15k237_cpg", 500, 2), ("fb15k237_plain", 1500, 3), ("wn18rr_cpg", 600, 4)])
def test_random_interleavings_of_the_pass_pipeline(workload, Q, seed):
    md = cdata.model_descriptors(workload, num_ent=2500)
    p = cdata.synthetic_params(md, 6)
    # (COPER_PIPELINE_SEQS: a longer soak of the same test -- profiles/r06_pipeline_soak.txt ran 4 x 5,000 sequences)
    n_stale = _run_sequences(md, p, Q, n_seq=int(os.environ.get("COPER_PIPELINE_SEQS", "230")), n_ops=24, seed=seed)
    assert n_stale > 0          # (the sequences do rewrite arrays behind their sorting: the guard was exercised, not just idle)


def test_random_interleavings_with_training_steps():
    md = dict(cdata._COMMON)
    md.update(_TRAIN_MD)
    md.update(batch_norm_train_stats=True, batch_norm_momentum=0.9, learning_rate=0.003, use_negative_sampling=True)
    p = cdata.synthetic_params(md, seed=8, ent_std=0.1)
    # (COPER_PIPELINE_TRAIN_SEQS: the same as a soak -- round 6's training step forks onto side streams of its own)
    _run_sequences(md, p, 400, n_seq=int(os.environ.get("COPER_PIPELINE_TRAIN_SEQS", "80")), n_ops=24, seed=int(os.environ.get("COPER_PIPELINE_TRAIN_SEED", "9")), train=True)
