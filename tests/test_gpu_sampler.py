"""GPU tests of the native negative samplers (`coper_sample_train_batch`): the construction rules of
CoPER_ConvE/qa_cpg/data.py:228-311 on every row, and -- the RNG stream of TensorFlow / NumPy cannot be reproduced -- the DISTRIBUTION the
reference's construction has: the sampled entities of a row are the head of a uniform permutation of all entities (distinct, every entity
equally likely at every position), the leading tails a uniform ordered subset of the record's tails."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _graph(rng, E, N, kmin, kmax):
    indptr, idx = [0], []
    for _ in range(N):
        k = int(rng.integers(kmin, kmax + 1))
        idx.extend(rng.choice(E, size=k, replace=False).tolist())      # (not sorted: the loader does not promise an order)
        indptr.append(len(idx))
    return dict(e1=rng.integers(0, E, N), rel=rng.integers(0, 7, N), tail_indptr=np.array(indptr, np.int64), tail_idx=np.array(idx, np.int64))


def _batches(ds, n):
    it = iter(ds)
    out = []
    for _ in range(n):
        b = next(it)
        torch.cuda.synchronize()
        out.append({k: v.cpu().numpy() for k, v in b.items()})
    return out


def _tails_of(s, e1, rel):
    """tail sets of the batch's rows: the records are found by (e1, rel) -- the graphs here give every record its own e1"""
    by_e1 = {}
    for i in range(len(s["e1"])):
        by_e1.setdefault((int(s["e1"][i]), int(s["rel"][i])), []).append(i)
    out = []
    for a, r in zip(e1, rel):
        recs = by_e1[(int(a), int(r))]
        t = set()
        for i in recs:
            t |= set(s["tail_idx"][s["tail_indptr"][i]:s["tail_indptr"][i + 1]].tolist())
        out.append(t)
    return out


@pytest.mark.parametrize("E,L", [(120, 40), (211, 37), (5000, 200), (14541, 1000), (40943, 1000), (300, 300), (70000, 2048)])
def test_one_positive_per_row_construction(E, L):
    """data.py:278-311: the row's positive in front, L - 1 DISTINCT sampled entities behind it, labels = membership in the record's tails.
    (E < 4 (L - 1): the Fisher-Yates path; otherwise the draw-and-skip-repeats path.)"""
    from coper_amd.data import DeviceTrainDataset
    rng = np.random.default_rng(E + L)
    N = 150
    s = _graph(rng, E, N, 1, 9)
    s["e1"] = np.arange(N) % E if N <= E else s["e1"]
    ds = DeviceTrainDataset(s, E, batch_size=64, num_labels=L, seed=3, device="cuda:0")
    assert ds.native
    for b in _batches(ds, 3):
        lk, lab = b["lookup_values"], b["e2_multi"]
        assert lk.shape == (64, L) and lk.dtype == np.int32 and lab.dtype == np.float32 and lk.min() >= 0 and lk.max() < E
        assert np.array_equal(lk[:, 0].astype(np.int64), b["e2"]) and (lab[:, 0] == 1).all()
        tails = _tails_of(s, b["e1"], b["rel"]) if N <= E else None
        for r in range(64):
            assert len(set(lk[r, 1:].tolist())) == L - 1                                   # distinct among themselves
            if tails is not None:
                assert int(lk[r, 0]) in tails[r]
                assert np.array_equal(lab[r], np.array([float(int(v) in tails[r]) for v in lk[r]], np.float32))


@pytest.mark.parametrize("E,L,prop,kmax", [(211, 37, 5.0, 11), (14541, 1000, 100.0, 40), (2000, 600, 10.0, 300), (9000, 64, 1.0, 5)])
def test_proportional_construction(E, L, prop, kmax):
    """data.py:228-277: `lead` tails of the record in front (all of them when there are at most int(L / (1 + prop)), that many otherwise),
    distinct; sampled entities behind, distinct; labels = membership; e2 = the first of the shuffled tails."""
    from coper_amd.data import DeviceTrainDataset
    rng = np.random.default_rng(7 * E + L)
    N = 90
    s = _graph(rng, E, N, 1, kmax)
    s["e1"] = np.arange(N)
    need = int(1.0 / (1.0 + prop) * L)
    ds = DeviceTrainDataset(s, E, batch_size=48, num_labels=L, seed=2, device="cuda:0", one_positive_label_per_sample=False, prop_negatives=prop)
    assert ds.native
    for b in _batches(ds, 4):
        for r in range(48):
            i = int(b["e1"][r])
            t = s["tail_idx"][s["tail_indptr"][i]:s["tail_indptr"][i + 1]].tolist()
            lk, lab = b["lookup_values"][r], b["e2_multi"][r]
            lead = len(t) if len(t) <= need else max(L - min(E, L - need), 0)
            assert set(lk[:lead].tolist()) <= set(t) and len(set(lk[:lead].tolist())) == lead and len(set(lk[lead:].tolist())) == L - lead
            assert np.array_equal(lab, np.array([float(int(v) in set(t)) for v in lk], np.float32))
            assert int(b["e2"][r]) in t and (lead == 0 or int(b["e2"][r]) == int(lk[0]))
            assert int(b["rel"][r]) == int(s["rel"][i])


def test_a_batch_is_a_function_of_seed_and_batch_number():
    from coper_amd.data import DeviceTrainDataset
    rng = np.random.default_rng(1)
    s = _graph(rng, 3000, 200, 1, 6)
    a = _batches(DeviceTrainDataset(s, 3000, 64, num_labels=100, seed=11, device="cuda:0"), 3)
    b = _batches(DeviceTrainDataset(s, 3000, 64, num_labels=100, seed=11, device="cuda:0"), 3)
    c = _batches(DeviceTrainDataset(s, 3000, 64, num_labels=100, seed=12, device="cuda:0"), 1)
    for x, y in zip(a, b):
        assert all(np.array_equal(x[k], y[k]) for k in x)
    assert not np.array_equal(a[0]["lookup_values"][:, 1:], a[1]["lookup_values"][:, 1:])      # another batch: other draws
    assert not np.array_equal(a[0]["lookup_values"][:, 1:], c[0]["lookup_values"][:, 1:])      # another seed too


@pytest.mark.parametrize("E,L", [(97, 60), (2000, 101)])
def test_sampled_entities_are_uniform(E, L):
    """Every entity equally likely at every sampled position (both paths: E < 4 n and E >= 4 n): chi-square of the entity counts over all
    sampled positions, of the counts at the FIRST sampled position, and the mean label rate against the graph's."""
    from coper_amd.data import DeviceTrainDataset
    rng = np.random.default_rng(5)
    s = _graph(rng, E, 64, 2, 2)
    ds = DeviceTrainDataset(s, E, batch_size=256, num_labels=L, seed=8, device="cuda:0")
    n_b = 40
    bs = _batches(ds, n_b)
    neg = np.concatenate([b["lookup_values"][:, 1:] for b in bs])                 # [rows, L - 1]
    for pos in (0, 1, (L - 1) // 2, L - 2):           # (positions of one row exclude one another: each position on its own, rows are independent)
        sample = neg[:, pos]
        cnt = np.bincount(sample, minlength=E).astype(np.float64)
        exp = len(sample) / E
        chi2 = ((cnt - exp) ** 2 / exp).sum()
        assert abs(chi2 - (E - 1)) < 6.0 * np.sqrt(2.0 * (E - 1)), (pos, chi2, E)
    assert (neg[:, 0] != neg[:, 1]).all()
    # the SECOND position given the first: uniform over the other E - 1 entities (difference mod E is uniform on 1 .. E - 1)
    diff = (neg[:, 1].astype(np.int64) - neg[:, 0]) % E
    cnt = np.bincount(diff, minlength=E)[1:].astype(np.float64)
    exp = len(diff) / (E - 1)
    assert abs(((cnt - exp) ** 2 / exp).sum() - (E - 2)) < 6.0 * np.sqrt(2.0 * (E - 2))
    lab = np.concatenate([b["e2_multi"][:, 1:] for b in bs])
    assert abs(lab.mean() - 2.0 / E) < 6.0 * np.sqrt((2.0 / E) / lab.size)        # two known tails per record: P(a sampled entity is one) = 2 / E


@pytest.mark.parametrize("n_tails", [12, 40])
def test_leading_tails_are_a_uniform_ordered_subset(n_tails):
    """Proportional sampler: a record with 12 (Fisher-Yates path) or 40 (draw-and-skip path) tails and room for 4: every tail equally likely
    in front (e2), every tail kept equally often."""
    from coper_amd.data import DeviceTrainDataset
    E, L, prop = 500, 24, 5.0                      # need = 4
    tails = np.arange(100, 100 + n_tails)
    s = dict(e1=np.array([3]), rel=np.array([1]), tail_indptr=np.array([0, n_tails]), tail_idx=tails)
    ds = DeviceTrainDataset(s, E, batch_size=512, num_labels=L, seed=4, device="cuda:0", one_positive_label_per_sample=False, prop_negatives=prop)
    bs = _batches(ds, 12)
    lead = np.concatenate([b["lookup_values"][:, :4] for b in bs])
    n = len(lead)
    assert all(len(set(r.tolist())) == 4 and set(r.tolist()) <= set(tails.tolist()) for r in lead[:200])
    first = np.bincount(lead[:, 0] - 100, minlength=n_tails)
    kept = np.bincount(lead.reshape(-1) - 100, minlength=n_tails)
    assert np.abs(first - n / n_tails).max() < 6.0 * np.sqrt(n / n_tails) and np.abs(kept - 4.0 * n / n_tails).max() < 6.0 * np.sqrt(4.0 * n / n_tails)


def test_tail_lists_longer_than_the_hash_set_and_native_agrees_with_the_torch_construction():
    """A record with 9,000 known tails (beyond the 8,192 the workgroup's set holds: its memberships are found by scanning the list), beside
    short ones; and the label rates of the native sampler against the torch-op construction on the same graph."""
    from coper_amd.data import DeviceTrainDataset
    rng = np.random.default_rng(9)
    E, L = 20000, 64
    big = rng.choice(E, size=9000, replace=False)
    s = dict(e1=np.array([0, 1, 2]), rel=np.array([0, 1, 2]), tail_indptr=np.array([0, 9000, 9003, 9004]),
             tail_idx=np.concatenate([big, rng.choice(E, 3, replace=False), rng.choice(E, 1)]).astype(np.int64))
    for one_pos in (True, False):
        rates = []
        for native in (True, False):
            ds = DeviceTrainDataset(s, E, batch_size=96, num_labels=L, seed=6, device="cuda:0", one_positive_label_per_sample=one_pos,
                                    prop_negatives=3.0, native=native)
            bs = _batches(ds, 6)
            pos = 0.0
            for b in bs:
                for r in range(96):
                    i = int(b["e1"][r])
                    t = set(s["tail_idx"][s["tail_indptr"][i]:s["tail_indptr"][i + 1]].tolist())
                    lk = b["lookup_values"][r]
                    assert np.array_equal(b["e2_multi"][r], np.array([float(int(v) in t) for v in lk], np.float32))
                pos += float(b["e2_multi"][b["e1"] == 0].mean()) if (b["e1"] == 0).any() else 0.0
            rates.append(pos / len(bs))
        assert abs(rates[0] - rates[1]) < 0.05, rates      # rows of the 9,000-tail record: ~0.45 of the sampled entities are known tails


def test_the_c_abi_refuses_what_it_cannot_do():
    from coper_amd import _lib
    lib = _lib.load()
    z = torch.zeros(16, dtype=torch.int64, device="cuda:0")
    o32 = torch.zeros(16, dtype=torch.int32, device="cuda:0")
    of = torch.zeros(16, dtype=torch.float32, device="cuda:0")
    args = lambda L, E, pos=True: (0, z.data_ptr(), z.data_ptr() if pos else None, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, L, E, 0, 10.0, 1,
                                   1, 0, z.data_ptr(), z.data_ptr(), z.data_ptr(), o32.data_ptr(), of.data_ptr(), None)
    assert lib.coper_sample_train_batch(*args(4096, 100000)) == 7        # COPER_EUNSUPPORTED: L beyond the LDS plan
    assert lib.coper_sample_train_batch(*args(8, 4)) == 1                # COPER_EINVAL: more labels than entities (data.py:146-147)
    assert lib.coper_sample_train_batch(*args(8, 100, pos=False)) == 1   # one positive per row needs the positives


def test_sampler_soak_short():
    """tests/sampler_soak.py: random graphs, label counts, batch sizes and samplers; every row by the construction rules."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "sampler_soak.py"), "120", "9"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "every row by the construction rules" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
