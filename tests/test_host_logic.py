"""CPU: host-side logic of the drop-in (batch contract, CSR filters, metric means, shard bounds)."""
import os

import numpy as np
import pytest

from coper_amd import data as cdata
from coper_amd.metrics import collect_batches, hits_and_means
from coper_amd.sharding import shard_bounds


def test_csr_dense_roundtrip_and_batches():
    md = cdata.model_descriptors("nations_cpg")
    loader = cdata.SyntheticKGLoader("nations_cpg", seed=1, queries=37)
    ds = loader.eval_dataset(batch_size=8, dense_mask=True)
    assert len(ds) == 5
    whole = ds.as_single_batch()
    n = 0
    for b in ds:
        B = len(b["e1"])
        assert b["e1"].dtype == np.int64 and b["rel"].dtype == np.int64 and b["e2"].dtype == np.int64
        assert b["lookup_values"].shape == (B, 0) and b["lookup_values"].dtype == np.int32      # data.py:205-213
        assert b["e2_multi"].shape == (B, md["num_ent"]) and b["e2_multi"].dtype == np.float32  # data.py:182-186
        ip, ix = cdata.dense_filter_to_csr(b["e2_multi"])
        assert np.array_equal(ip, b["filt_indptr"]) and np.array_equal(ix, b["filt_idx"])
        assert all(b["e2_multi"][i, b["e2"][i]] == 1.0 for i in range(B))   # the target is a known answer
        assert np.all(b["rel"] < md["num_rel"] // 2)                         # forward relations only
        n += B
    assert n == 37
    # draining batches (with dense masks only) reproduces the single-batch CSR
    dense_only = [dict(e1=b["e1"], e2=b["e2"], rel=b["rel"], e2_multi=b["e2_multi"]) for b in ds]
    got = collect_batches(dense_only)
    for k in ("e1", "e2", "rel", "filt_indptr", "filt_idx"):
        assert np.array_equal(got[k], whole[k]), k
    got2 = collect_batches(iter(ds))
    assert np.array_equal(got2["filt_idx"], whole["filt_idx"])


def test_empty_and_ragged_batches():
    got = collect_batches([])
    assert len(got["e1"]) == 0 and got["filt_indptr"].tolist() == [0]
    b0 = dict(e1=np.zeros(0, np.int64), e2=np.zeros(0, np.int64), rel=np.zeros(0, np.int64),
              filt_indptr=np.zeros(1, np.int64), filt_idx=np.zeros(0, np.int64))
    b1 = dict(e1=np.array([1, 2]), e2=np.array([3, 4]), rel=np.array([0, 1]), filt_indptr=np.array([0, 0, 2]),
              filt_idx=np.array([4, 9]))
    got = collect_batches([b0, b1, b0])
    assert got["filt_indptr"].tolist() == [0, 0, 2] and got["filt_idx"].tolist() == [4, 9]
    mr, mrr, hits = hits_and_means([], (1, 10))
    assert np.isnan(mr) and np.isnan(mrr) and np.isnan(hits[1])


def test_hits_and_means_reference_arithmetic():
    ranks = [1, 2, 3, 11, 21]
    mr, mrr, hits = hits_and_means(ranks)
    assert mr == np.mean(ranks) and mrr == np.mean(1. / np.array(ranks))
    assert hits == {1: 0.2, 3: 0.6, 5: 0.6, 10: 0.6, 20: 0.8}
    assert set(hits) == {1, 3, 5, 10, 20}          # default hits_to_compute (metrics.py:23)


@pytest.mark.parametrize("n,w", [(14, 1), (14, 4), (14541, 8), (10_000_000, 8), (5, 8)])
def test_shard_bounds_partition(n, w):
    cuts = [shard_bounds(n, w, r) for r in range(w)]
    assert cuts[0][0] == 0 and cuts[-1][1] == n
    assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
    sizes = [hi - lo for lo, hi in cuts]
    assert max(sizes) - min(sizes) <= 1


def test_synthetic_params_are_seeded_and_complete():
    md = cdata.model_descriptors("nations_cpg")
    a, b = cdata.synthetic_params(md, 3), cdata.synthetic_params(md, 3)
    assert set(a) == set(cdata.param_shapes(md))
    assert all(np.array_equal(a[k], b[k]) and a[k].dtype == np.float32 for k in a)
    c = cdata.synthetic_params(md, 4)
    assert not np.array_equal(a["ent_emb"], c["ent_emb"])
    q = cdata.synthetic_queries(md, 50, 0)
    for i in range(50):
        row = q["filt_idx"][q["filt_indptr"][i]:q["filt_indptr"][i + 1]]
        assert np.all(np.diff(row) > 0) and q["e2"][i] in row       # sorted unique, contains the target


def test_weight_import_helpers(tmp_path):
    import pickle
    from coper_amd.weights import leaf_name, load_best_embeddings_pickle, load_npz
    assert leaf_name("variables/variables/fc_weights/CPG/Projection0:0") == "fc_weights/CPG/Projection0"
    assert leaf_name("variables/Conv1BN/moving_variance:0") == "Conv1BN/moving_variance"
    rel, ent = np.ones((4, 3), np.float64), np.zeros((5, 2), np.float64)
    with open(tmp_path / "best_embeddings.ckpt", "wb") as f:           # run_cpg.py:242-249
        pickle.dump([rel, ent], f)
    got = load_best_embeddings_pickle(tmp_path / "best_embeddings.ckpt")
    assert got["rel_emb"].dtype == np.float32 and got["rel_emb"].shape == (4, 3) and got["ent_emb"].shape == (5, 2)
    with open(tmp_path / "e.ckpt", "wb") as f:
        pickle.dump(ent, f)
    assert set(load_best_embeddings_pickle(tmp_path / "e.ckpt", is_parameter_lookup=True)) == {"ent_emb"}
    np.savez(tmp_path / "w.npz", **{"variables/variables/pred_bias:0": np.arange(3.0)})
    assert list(load_npz(tmp_path / "w.npz")) == ["pred_bias"]


def test_bench_finds_the_committed_pmc_summaries():
    """bench.py replays HBM bytes per launch and the matrix-pipe busy fraction from the committed rocprofv3 summaries
    (profiles/): the kernels it names must be the ones the summaries hold, with their provenance."""
    import bench
    assert bench.score_kernel_name("bf16x3", 200) == "k_score_count3_bf16x3" and bench.score_kernel_name("bf16x3", 32) == "k_score_count3_bf16x3"
    assert bench.score_kernel_name("f32", 200) == "k_score_count_f32"
    e = {}
    bench.pmc_traffic(e, "fb15k237_cpg", 20480, "coper::k_score_count3_bf16x3")
    assert e["traffic"] and e["traffic"] < 1e9 and e["traffic_source"].startswith("profiles/")
    bench.pmc_mfma_busy(e, "fb15k237_cpg", 20480, "coper::k_score_count3_bf16x3")
    assert 0.5 < e["pmc"]["mfma_busy_frac"] < 1.0 and "profiles/" in e["pmc"]["source"]
    d = {}
    bench.pmc_traffic(d, "fb15k237_cpg", 20480, "coper::k_dense_fused_bf16x3")
    assert 0.9e9 < d["traffic"] < 1.2e9
    h = {}
    bench.pmc_traffic(h, "synth10m_cpg", 128, "coper::k_score_count3_bf16x3", exact="coper::k_score_count3_bf16x3<8, 0, 2, 0>")
    assert 1.0e10 < h["traffic"] < 1.15e10        # the 10M x 256 table read once: PMC bytes within 1.1x of the algorithmic 10.28 GB
    assert "r06" in h["traffic_source"]           # (the newest committed collection)
    for wl, Q in (("fb15k237_plain", 20480), ("wn18rr_cpg", 3072)):      # round 5: counter traffic of the encoder on the other configs too
        t = {}
        bench.pmc_traffic(t, wl, Q, "coper::k_dense_fused_bf16x3")
        assert t["traffic"] and t["traffic"] < 0.3e9, (wl, t)
    t = {}
    bench.pmc_traffic(t, "synth10m_cpg", 4096, "coper::k_score_count3_bf16x3", exact="coper::k_score_count3_bf16x3<8, 0, 2, 2>")
    # the top-k launch on the WHOLE table (round 6: summaries are keyed by grid size -- round 5's 8.54 GB was the mean of this launch and
    # the 1/8-shard launch of scale.projected): the 10 GB image through L2 / MALL for 32 query tiles + 2.6 GB of 64-entity maxima written
    assert 1.4e10 < t["traffic"] < 1.6e10
    none = {}
    bench.pmc_traffic(none, "no_such_workload", 1, "coper::k_nothing")
    assert none["traffic"] is None and "traffic_source" not in none


def test_unsorted_filter_rows_are_sorted_before_the_kernels_see_them():
    """include/coper_hip.h: filter rows sorted ascending.  Batches with hand-built, unsorted rows go through canonical_csr."""
    from coper_amd.data import canonical_csr
    from coper_amd.metrics import collect_batches
    indptr = np.array([0, 3, 3, 7, 8])
    idx = np.array([5, 2, 9, 4, 4, 1, 8, 0])
    ip, ix = canonical_csr(indptr, idx)
    assert ip.tolist() == indptr.tolist() and ix.tolist() == [2, 5, 9, 1, 4, 4, 8, 0]
    ok = np.array([2, 5, 9, 1, 4, 4, 8, 0])
    ip2, ix2 = canonical_csr(indptr, ok)
    assert ix2 is not None and ix2.tolist() == ok.tolist()
    assert canonical_csr(np.array([0, 0]), np.zeros(0, np.int64))[1].tolist() == []
    q = collect_batches([dict(e1=[0, 1], e2=[5, 4], rel=[0, 0], filt_indptr=[0, 3, 5], filt_idx=[9, 5, 2, 4, 1])])
    assert q["filt_idx"].tolist() == [2, 5, 9, 1, 4] and q["filt_indptr"].tolist() == [0, 3, 5]


def test_one_fused_call_per_chunk_whatever_the_filter_rows_hold():
    """Round 2 routed passes with thousands of known answers per 32-query block to the two-call path in Python; since round 3
    coper_encode_rank deals such blocks over the chip itself, so local_rank_pass always makes the one call."""
    from coper_amd import sharding
    calls = []

    class M:
        def rank_pass(self, *a, **k):
            calls.append("fused"); return "r", None
        def encode(self, e1, rel):
            calls.append("encode"); return "h"
        def rank(self, h, e2, ip, ix, filt_nnz=None):
            calls.append("rank"); return "r", "ne"

    class TwoCall:
        encode, rank = M.encode, M.rank

    light = np.arange(0, 5 * 101, 5)                       # 100 queries, 5 entries each
    heavy = light.copy(); heavy[40:] += 3000               # one query with 3,000 known answers
    z = np.zeros(100, np.int64)
    sharding.local_rank_pass(M(), dict(e1=z, rel=z, e2=z, filt_indptr=light, filt_idx=np.zeros(light[-1], np.int64)))
    sharding.local_rank_pass(M(), dict(e1=z, rel=z, e2=z, filt_indptr=heavy, filt_idx=np.zeros(heavy[-1], np.int64)))
    sharding.local_rank_pass(TwoCall(), dict(e1=z, rel=z, e2=z, filt_indptr=heavy, filt_idx=np.zeros(heavy[-1], np.int64)))
    assert calls == ["fused", "fused", "encode", "rank"]
    assert not hasattr(sharding, "_heavy_filter_rows")


def test_bench_self_launch_propagates_a_failing_rank():
    """`python bench.py --gpus 2` from a plain shell starts its ranks itself; on this box (no GPU) both ranks fail at
    device set-up, the launcher stops what is left, prints no JSON line and exits non-zero."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "1",
                          "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "stopping the other ranks" in out.stderr and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_generated_asm_regions_are_up_to_date(tmp_path):
    """coper_amd/csrc/sc3_region_asm{,_gm,_gm64}.inc are generated (tools/gen_sc3_region_asm.py) and committed: the committed files must be
    what the generator writes today."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, os.path.join(root, "tools", "gen_sc3_region_asm.py"), str(tmp_path)], check=True, capture_output=True)
    for name in ("sc3_region_asm.inc", "sc3_region_asm_gm.inc", "sc3_region_asm_gm64.inc"):
        assert open(os.path.join(str(tmp_path), name)).read() == open(os.path.join(root, "coper_amd", "csrc", name)).read(), name


def test_hits_means_native_equals_numpy_bit_for_bit():
    """coper_hits_means (host code of the library; metrics.hits_and_means takes it for int32 ranks): mean rank, MRR and Hits@k are
    the float64 values the reference's NumPy expressions give (metrics.py:53-57, 65-76) -- np.mean's summation order included:
    pairwise inside chunks of np.getbufsize() elements -- for sizes around every boundary of that order; ranks below 1 are refused
    and take the NumPy route."""
    import ctypes as C
    from coper_amd import _lib, metrics
    lib = _lib.load()
    rng = np.random.default_rng(3)
    levels = (1, 3, 5, 10, 20)
    for n in (1, 7, 8, 9, 127, 128, 129, 1023, 8191, 8192, 8193, 16384, 20480, 70001):
        for top in (3, 14542, 5_000_000, 40_000_000):
            r = rng.integers(1, top, n).astype(np.int32)
            mr, mrr, hits = metrics.hits_and_means(r, levels)
            r64 = r.astype(np.int64)
            assert mr == np.mean(r64) and mrr == np.mean(1. / r64), (n, top)
            for k in levels:
                assert hits[k] == np.mean(np.where(r64 <= k, 1.0, 0.0))
            assert (mr, mrr, hits) == metrics.hits_and_means(r64, levels)          # the NumPy route of the same function
    r = rng.integers(1, 100, 50).astype(np.int32)
    mr, mrr = C.c_double(), C.c_double()
    lv = (C.c_int32 * 2)(1, 10)
    hv = (C.c_double * 2)()
    assert lib.coper_hits_means(C.c_void_p(r.ctypes.data), 50, lv, 2, C.byref(mr), C.byref(mrr), hv) == 0
    assert hv[1] == np.count_nonzero(r <= 10) / 50
    r[3] = 0
    assert lib.coper_hits_means(C.c_void_p(r.ctypes.data), 50, lv, 2, C.byref(mr), C.byref(mrr), hv) == 1
    assert lib.coper_hits_means(C.c_void_p(r.ctypes.data), 0, lv, 2, C.byref(mr), C.byref(mrr), hv) == 1
    with np.errstate(divide="ignore"):
        assert metrics.hits_and_means(r, levels)[1] == np.inf                      # (what np.mean(1. / ranks) gives for a rank of 0)


def test_pack_ids_i32_checks_range_and_row_order():
    """coper_pack_ids_i32 (host code): the narrowing copy into the staging buffer reports ids beyond int32 and filter rows that
    are not ascending -- what `canonical_csr` and `stage_batch` establish with NumPy passes -- and nothing else."""
    import ctypes as C
    from coper_amd import _lib
    from coper_amd.data import canonical_csr
    lib = _lib.load()
    rng = np.random.default_rng(5)

    def pack(a, ip=None):
        a = np.ascontiguousarray(a, np.int64)
        out = np.full(a.size, -7, np.int32)
        st = C.c_int32(-1)
        rc = lib.coper_pack_ids_i32(C.c_void_p(a.ctypes.data), a.size, C.c_void_p(out.ctypes.data),
                                    C.c_void_p(ip.ctypes.data) if ip is not None else None, len(ip) - 1 if ip is not None else 0, C.byref(st))
        return rc, st.value, out

    for trial in range(200):
        B = int(rng.integers(1, 40))
        lens = rng.integers(0, 9, B)
        ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        rows = [np.sort(rng.integers(0, 50, n)) for n in lens]
        if trial % 3 == 1 and lens.max() >= 2:                       # one row out of order somewhere
            r = int(np.argmax(lens))
            rows[r] = rows[r][::-1].copy()
            if np.all(rows[r][1:] >= rows[r][:-1]):
                rows[r][0] = rows[r][-1] + 1
        ix = np.concatenate(rows).astype(np.int64) if ip[-1] else np.zeros(0, np.int64)
        rc, st, out = pack(ix, ip)
        assert rc == 0 and np.array_equal(out, ix.astype(np.int32))
        _, fixed = canonical_csr(ip, ix)
        assert bool(st & 2) == (not np.array_equal(fixed, ix)), (trial, st)
        assert not (st & 1)
    big = np.array([1, 2 ** 31 - 1, -2 ** 31, 5], np.int64)
    assert pack(big)[:2] == (0, 0)
    for v in (2 ** 31, -2 ** 31 - 1, 2 ** 40, -2 ** 62):
        b2 = big.copy(); b2[2] = v
        assert pack(b2)[:2] == (0, 1), v
    # malformed row pointers are refused, not read
    ix = np.arange(6, dtype=np.int64)
    for bad_ip in ([0, 3, 5], [1, 3, 6], [0, 7, 6], [0, 4, 3, 6]):
        assert pack(ix, np.asarray(bad_ip, np.int64))[0] == 1, bad_ip
    assert pack(np.zeros(0, np.int64), np.zeros(4, np.int64))[:2] == (0, 0)      # empty rows only


def test_entity_sharded_plan_and_the_emulated_rank(oracle_chain):
    """coper_amd.sharding (round 6): the host plan of a chunk -- owners, slots, the relation split, all in one buffer -- against a
    plain restatement, ids outside the table refused; and `emulate_world` (bench.py's scale.projected: ONE rank's work of a G-rank
    job, every all-gather a local copy) runs the whole exchange without a process group."""
    import numpy as np
    import torch
    from coper_amd import data as cdata
    from coper_amd.sharding import EntityShardedRanker, shard_bounds
    from tests.oracle_scorer import OracleShardScorer
    md = dict(cdata._COMMON, num_ent=203, num_rel=12, ent_emb_size=32, rel_emb_size=8, emb_h=4, emb_w=8, conv_num_channels=4,
              context_rel_conv=None, context_rel_out=[])
    p = cdata.synthetic_params(md, seed=11)
    G, g = 4, 1
    sc = OracleShardScorer(p, md, shard_bounds(md["num_ent"], G, g))
    er = EntityShardedRanker(sc, emulate_world=(G, g))
    assert er.world == G and er.rank_id == g and not er.overlap
    q = cdata.synthetic_queries(md, 37, seed=3, mean_filter=2.0, max_filter=8)
    pl = er.plan(q)
    ids = np.concatenate([q["e1"], q["e2"]])
    bounds = [shard_bounds(md["num_ent"], G, r) for r in range(G)]
    owner = np.array([next(r for r, (lo, hi) in enumerate(bounds) if lo <= i < hi) for i in ids])
    counts = np.bincount(owner, minlength=G)
    assert pl.cap1 == counts.max() and pl.n_mine == counts[g]
    assert np.array_equal(pl.mine_ids, ids[owner == g])                                   # stable: in batch order
    take = np.concatenate([er._dev(pl, "take1").numpy(), er._dev(pl, "take2").numpy()])
    seen = np.zeros(G, int)
    for pos, o in enumerate(owner):                                                       # position p sits behind its owner's header row, in order
        assert take[pos] == o * (pl.cap1 + 1) + 1 + seen[o]
        seen[o] += 1
    rel = np.asarray(q["rel"])
    assert np.array_equal(er._dev(pl, "sel").numpy(), np.nonzero(rel % G == g)[0]) and pl.n_enc == int((rel % G == g).sum())
    assert pl.cap2 == np.bincount(rel % G, minlength=G).max()
    for key, bad in (("e1", -1), ("e2", md["num_ent"])):
        qb = dict(q)
        qb[key] = np.where(np.arange(len(q[key])) == 5, bad, q[key])
        with pytest.raises(ValueError):
            er.plan(qb)
    out = list(er.rank_stream([q, cdata.synthetic_queries(md, 20, seed=4, mean_filter=2.0, max_filter=8)], k=3, window=2))
    assert len(out) == 2 and out[0][0].shape == (37,) and out[1][2].shape == (20, 3)
