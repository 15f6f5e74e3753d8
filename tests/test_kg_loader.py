"""CPU: the TSV loader (coper_amd/kg_loader.py) against the outputs of the REFERENCE'S OWN loader code
(qa_cpg/data.py load_and_preprocess / _write_graph / _assign_ids, run under a stub tensorflow module by
oracle/gen_golden.py) on a split of the nell-995 dev triples the reference ships."""
import json
import os
import shutil

import numpy as np
import pytest

from coper_amd.kg_loader import TSVKGLoader


def _ref_samples(path):
    out = []
    for line in open(path):
        s = json.loads(line)
        out.append((s["e1"], s["rel"], s["e2"], frozenset(t for t in s["e2_multi"].split(" ") if t)))
    return sorted(out, key=lambda x: (x[0], x[1], x[2]))


def _my_samples(samples):
    return sorted(((s["e1"], s["rel"], s["e2"], frozenset(s["e2_multi"])) for s in samples), key=lambda x: (x[0], x[1], x[2]))


@pytest.mark.parametrize("clean", [False, True])
def test_json_samples_equal_reference_loader(golden_dir, tmp_path, clean):
    ref_dir = os.path.join(golden_dir, "kg_ref_clean" if clean else "kg_ref")
    for f in ("train.txt", "dev.txt", "test.txt"):
        shutil.copy(os.path.join(golden_dir, "kg_tsv", f), tmp_path)
    loader = TSVKGLoader(str(tmp_path), "nell-995-test", needs_test_set_cleaning=clean)
    samples = loader.load_and_preprocess(write_json=True)
    for split in ("train", "dev", "test", "full"):
        assert _my_samples(samples[split]) == _ref_samples(os.path.join(ref_dir, "e1rel_to_e2_%s.json" % split)), split
        # the JSON files written here parse to the same samples as the reference's files
        assert _ref_samples(str(tmp_path / ("e1rel_to_e2_%s.json" % split))) == _ref_samples(os.path.join(ref_dir, "e1rel_to_e2_%s.json" % split))


@pytest.mark.parametrize("clean", [False, True])
def test_ids_and_eval_batches_follow_reference_id_files(golden_dir, tmp_path, clean):
    ref_dir = os.path.join(golden_dir, "kg_ref_clean" if clean else "kg_ref")
    for f in ("train.txt", "dev.txt", "test.txt"):
        shutil.copy(os.path.join(golden_dir, "kg_tsv", f), tmp_path)
    for f in ("entities.txt", "relations.txt"):          # id files written by the reference define the ids
        shutil.copy(os.path.join(ref_dir, f), tmp_path)
    loader = TSVKGLoader(str(tmp_path), "nell-995-test", needs_test_set_cleaning=clean)
    ent, rel = loader.assign_ids()
    ref_ent = {l.strip(): i for i, l in enumerate(open(os.path.join(ref_dir, "entities.txt")))}
    ref_rel = {l.strip(): i for i, l in enumerate(open(os.path.join(ref_dir, "relations.txt")))}
    assert ent == ref_ent and rel == ref_rel
    assert loader.num_ent == len(ref_ent) and loader.num_rel == len(ref_rel)
    # num_rel counts the `_reverse` relations (data.py:422-428,338)
    assert sum(1 for r in ref_rel if r.endswith("_reverse")) * 2 == len(ref_rel)
    for split in ("dev", "test"):
        enc = loader.encoded_split(split)
        # expected: reference JSON lines of the split, forward relations only (run_cpg.py:156), through the id maps
        exp = []
        for e1, r, e2, tails in _ref_samples(os.path.join(ref_dir, "e1rel_to_e2_%s.json" % split)):
            if r.endswith("_reverse"):
                continue
            exp.append((ref_ent[e1], ref_rel[r], ref_ent[e2], frozenset(ref_ent[t] for t in tails)))
        got = []
        for i in range(len(enc["e1"])):
            row = enc["filt_idx"][enc["filt_indptr"][i]:enc["filt_indptr"][i + 1]]
            assert np.all(np.diff(row) > 0)                  # CSR rows sorted unique
            assert enc["e2"][i] in row                       # the target is a known answer (full-graph labels)
            got.append((int(enc["e1"][i]), int(enc["rel"][i]), int(enc["e2"][i]), frozenset(int(v) for v in row)))
        assert sorted(got, key=lambda x: x[:3]) == sorted(exp, key=lambda x: x[:3]), split
    ds = loader.eval_dataset(dataset_type="test", batch_size=16, dense_mask=True)
    n = 0
    for b in ds:
        assert b["e2_multi"].shape == (len(b["e1"]), loader.num_ent)
        n += len(b["e1"])
    assert n == len(loader.encoded_split("test")["e1"])


def test_ids_are_deterministic_without_id_files(golden_dir, tmp_path):
    for f in ("train.txt", "dev.txt", "test.txt"):
        shutil.copy(os.path.join(golden_dir, "kg_tsv", f), tmp_path)
    a = TSVKGLoader(str(tmp_path)).assign_ids(write_files=True)
    b = TSVKGLoader(str(tmp_path)).assign_ids()            # now read back from the files
    assert a == b and list(a[0]) == sorted(a[0])


def test_train_dataset_sampler_rules(tmp_path):
    """TrainDataset restates data.py:228-311: row construction rules of both samplers."""
    from coper_amd.data import TrainDataset
    rng = np.random.default_rng(0)
    E, N = 97, 40
    indptr = [0]
    idx = []
    for i in range(N):
        k = int(rng.integers(1, 9))
        idx.extend(sorted(rng.choice(E, size=k, replace=False)))
        indptr.append(len(idx))
    s = dict(e1=rng.integers(0, E, N), rel=rng.integers(0, 6, N), tail_indptr=np.array(indptr), tail_idx=np.array(idx))
    tails_of = {(int(s["e1"][i]), int(s["rel"][i])): set() for i in range(N)}
    for i in range(N):
        tails_of[(int(s["e1"][i]), int(s["rel"][i]))] |= set(idx[indptr[i]:indptr[i + 1]])
    # one positive per row (the default)
    ds = TrainDataset(s, E, batch_size=32, num_labels=20, seed=1)
    it = iter(ds)
    for _ in range(5):
        b = next(it)
        assert b["lookup_values"].shape == (32, 20) and b["lookup_values"].dtype == np.int32
        assert b["e2_multi"].shape == (32, 20) and b["e2_multi"].dtype == np.float32
        assert np.array_equal(b["lookup_values"][:, 0], b["e2"])            # the positive comes first (data.py:296)
        assert (b["e2_multi"][:, 0] == 1.0).all()
        for r in range(32):
            # labels = membership in the record's tail list; negatives are a run of distinct entities
            rec = [set(idx[indptr[i]:indptr[i + 1]]) for i in range(N)
                   if s["e1"][i] == b["e1"][r] and s["rel"][i] == b["rel"][r] and int(b["e2"][r]) in idx[indptr[i]:indptr[i + 1]]]
            assert any(np.array_equal(b["e2_multi"][r], np.array([float(v in t) for v in b["lookup_values"][r]], np.float32)) for t in rec)
            assert len(set(b["lookup_values"][r, 1:].tolist())) == 19
    # proportional sampler (data.py:228-277)
    ds2 = TrainDataset(s, E, batch_size=16, num_labels=22, one_positive_label_per_sample=False, prop_negatives=10.0, seed=2)
    b = next(iter(ds2))
    need = int(1.0 / 11.0 * 22)
    for r in range(16):
        lab, lk = b["e2_multi"][r], b["lookup_values"][r]
        recs = [idx[indptr[i]:indptr[i + 1]] for i in range(N) if s["e1"][i] == b["e1"][r] and s["rel"][i] == b["rel"][r]]
        ok = False
        for t in recs:
            npos = len(t)
            lead = npos if npos <= need else max(22 - min(E, 22 - need), 0)
            if set(lk[:lead].tolist()) <= set(t) and np.array_equal(lab, np.array([float(v in t) for v in lk], np.float32)):
                ok = True
        assert ok
    with pytest.raises(ValueError):
        TrainDataset(s, E, 4, num_labels=E + 1)


def test_tfrecord_files_round_trip(tmp_path, golden_dir):
    """The reference's TFRecord files (data.py:353-397, 574-594) without TensorFlow: framing known answers, Example
    round trip incl. negative ids and empty lists, and the loader's splits written and read back == the in-memory
    id arrays the engine feeds from."""
    import shutil
    import struct
    from coper_amd import tf_records as tr
    from coper_amd.kg_loader import TSVKGLoader
    # framing: length | masked crc(length) | data | masked crc(data)
    p = tmp_path / "x.tfrecords"
    assert tr.write_records(str(p), [b"", b"abc"]) == 2
    raw = p.read_bytes()
    assert raw[:8] == struct.pack("<Q", 0) and len(raw) == 12 + 4 + 12 + 3 + 4
    assert list(tr.read_records(str(p))) == [b"", b"abc"]
    bad = bytearray(raw); bad[-6] ^= 1
    p.write_bytes(bytes(bad))
    with pytest.raises(ValueError, match="corrupt"):
        list(tr.read_records(str(p)))
    ex = dict(e1=[7], e2=[-1], rel=[3], e2_multi=[], is_inverse=[1])
    assert tr.parse_example(tr.encode_example(ex)) == dict(e1=[7], e2=[-1], rel=[3], e2_multi=[], is_inverse=[1])
    big = dict(e2_multi=list(range(0, 300000, 997)), e1=[2 ** 40])
    assert tr.parse_example(tr.encode_example(big)) == big
    # a hand-assembled Example with an UNPACKED int64 list (legal wire form): features{feature{key:"e1" value{int64_list{value:5 value:6}}}}
    il = bytes([0x08, 5, 0x08, 6])
    feat = bytes([0x1A, len(il)]) + il
    entry = bytes([0x0A, 2]) + b"e1" + bytes([0x12, len(feat)]) + feat
    feats = bytes([0x0A, len(entry)]) + entry
    assert tr.parse_example(bytes([0x0A, len(feats)]) + feats) == {"e1": [5, 6]}
    # the loader's splits
    for f in ("train.txt", "dev.txt", "test.txt"):
        shutil.copy(os.path.join(golden_dir, "kg_tsv", f), tmp_path)
    loader = TSVKGLoader(str(tmp_path), "nell-995-test")
    files = loader.maybe_create_tf_record_files(str(tmp_path), write_tfrecords=True, max_records_per_file=500)
    assert sorted(files) == ["dev", "test", "train"] and len(files["train"]) > 1
    for split in ("dev", "test"):
        for inv in (False, True):
            got = tr.read_split(str(tmp_path), split, include_inv_relations=inv)
            want = loader.encoded_split(split, include_inv_relations=inv)
            for k in ("e1", "e2", "rel", "filt_indptr", "filt_idx"):
                assert np.array_equal(got[k], want[k]), (split, inv, k)
    tr_got = tr.read_split(str(tmp_path), "train", include_inv_relations=True)
    tr_want = loader.train_samples(include_inv_relations=True)
    nonempty = np.diff(tr_got["filt_indptr"]) > 0
    assert np.array_equal(tr_got["e1"][nonempty], tr_want["e1"]) and np.array_equal(tr_got["rel"][nonempty], tr_want["rel"])
    assert np.array_equal(tr_got["filt_idx"], tr_want["tail_idx"])
    # existing files are not recreated (data.py:360-363)
    again = loader.maybe_create_tf_record_files(str(tmp_path), write_tfrecords=True)
    assert again == {k: sorted(v) for k, v in files.items()}


def test_tfrecord_loader_equals_tsv_loader(tmp_path, golden_dir):
    """TFRecordKGLoader on a directory holding only <split>-*.tfrecords + id files == TSVKGLoader on the TSVs."""
    import shutil
    from coper_amd.kg_loader import TFRecordKGLoader, TSVKGLoader
    src = tmp_path / "src"; src.mkdir()
    for f in ("train.txt", "dev.txt", "test.txt"):
        shutil.copy(os.path.join(golden_dir, "kg_tsv", f), src)
    a = TSVKGLoader(str(src), "nell-995-test")
    a.assign_ids(write_files=True)
    a.maybe_create_tf_record_files(str(src), write_tfrecords=True)
    rec = tmp_path / "rec"; rec.mkdir()
    for f in os.listdir(src):
        if f.endswith(".tfrecords") or f in ("entities.txt", "relations.txt"):
            shutil.copy(src / f, rec)
    b = TFRecordKGLoader(str(rec))
    b.maybe_create_tf_record_files()
    assert (b.num_ent, b.num_rel) == (a.num_ent, a.num_rel)
    for split in ("dev", "test"):
        qa, qb = a.eval_dataset(None, split).as_single_batch(), b.eval_dataset(None, split).as_single_batch()
        for k in ("e1", "e2", "rel", "filt_indptr", "filt_idx"):
            assert np.array_equal(qa[k], qb[k]), (split, k)
    ta, tb = a.train_samples(), None
    it = iter(b.train_dataset(None, batch_size=16, num_labels=20, seed=1))
    batch = next(it)
    assert batch["lookup_values"].shape == (16, 20) and batch["e2_multi"].shape == (16, 20) and batch["e2_multi"][:, 0].all()
    assert len(ta["e1"]) == len(b.train_dataset(None, batch_size=16, num_labels=20).s["e1"])


def test_tfrecord_reader_on_a_hand_assembled_file(tmp_path, golden_dir):
    """Bytes assembled by tests/golden/make_tfrecord_handmade.py (its own CRC, varints and protobuf framing; nothing of
    coper_amd imported): the reader is not tested against its own writer only.  Still a restatement of the published
    format, not a TensorFlow-written file (none can be produced here)."""
    import json
    from coper_amd import tf_records as R
    g = json.load(open(os.path.join(golden_dir, "tfrecord_handmade.json")))
    blob = bytes.fromhex(g["file_hex"])
    path = str(tmp_path / "test-0.tfrecords")
    with open(path, "wb") as f:
        f.write(blob)
    recs = list(R.read_records(path, verify=True))
    assert len(recs) == len(g["samples"])
    for rec, want in zip(recs, g["samples"]):
        assert R.parse_example(rec) == want
    # the framing words of the first record, as the generator computed them with its bitwise CRC
    assert int.from_bytes(blob[:8], "little") == g["first_length"] == len(recs[0])
    assert int.from_bytes(blob[8:12], "little") == g["first_length_crc"]
    assert int.from_bytes(blob[12 + len(recs[0]):16 + len(recs[0])], "little") == g["first_data_crc"]
    # samples in sorted-key packed form are also what the writer emits: byte-identical records
    assert R.encode_example(g["samples"][0]) == recs[0]
    assert R.encode_example(g["samples"][2]) == recs[2]
    # split view: the inverse sample (is_inverse = 1) and the one flagged -1 are dropped unless asked for
    s = R.read_split(str(tmp_path), "test")
    assert s["e1"].tolist() == [3, 14540] and s["rel"].tolist() == [2, 473]
    assert s["filt_indptr"].tolist() == [0, 3, 73]
    assert s["filt_idx"][:3].tolist() == [4, 17, 129]
    s = R.read_split(str(tmp_path), "test", include_inv_relations=True)
    assert s["e1"].tolist() == [3, 0, 14540, 5] and s["filt_indptr"].tolist() == [0, 3, 4, 74, 74]
    # a flipped byte is caught by the data CRC
    bad = bytearray(blob); bad[20] ^= 1
    with open(path, "wb") as f:
        f.write(bytes(bad))
    with pytest.raises(ValueError):
        list(R.read_records(path, verify=True))


def test_device_train_dataset_sampler_rules():
    """DeviceTrainDataset (here on torch's CPU device; the GPU suite runs it on the HIP device): the construction rules of
    the one-positive-per-row sampler (data.py:278-311) -- positive first, L - 1 distinct negatives, labels = membership in
    the record's tail list -- and the row stream: over an epoch every (record, tail) row appears, rows pass a shuffle buffer."""
    from coper_amd.data import DeviceTrainDataset
    rng = np.random.default_rng(3)
    E, N = 97, 40
    indptr, idx = [0], []
    for i in range(N):
        k = int(rng.integers(0, 9))          # records without tails produce no rows
        idx.extend(sorted(rng.choice(E, size=k, replace=False)))
        indptr.append(len(idx))
    s = dict(e1=rng.integers(0, E, N), rel=rng.integers(0, 6, N), tail_indptr=np.array(indptr), tail_idx=np.array(idx))
    ds = DeviceTrainDataset(s, E, batch_size=32, num_labels=20, seed=1, device="cpu", shuffle_buffer=40)
    it = iter(ds)
    seen = set()
    n_rows = len(idx)
    for _ in range(8 * (n_rows // 32 + 1)):
        b = {k: v.numpy() for k, v in next(it).items()}
        assert b["lookup_values"].shape == (32, 20) and b["lookup_values"].dtype == np.int32
        assert b["e2_multi"].shape == (32, 20) and b["e2_multi"].dtype == np.float32
        assert np.array_equal(b["lookup_values"][:, 0], b["e2"]) and (b["e2_multi"][:, 0] == 1.0).all()
        assert b["lookup_values"].min() >= 0 and b["lookup_values"].max() < E
        for r in range(32):
            recs = [i for i in range(N) if s["e1"][i] == b["e1"][r] and s["rel"][i] == b["rel"][r]
                    and int(b["e2"][r]) in idx[indptr[i]:indptr[i + 1]]]
            assert recs
            assert any(np.array_equal(b["e2_multi"][r], np.array([float(v in set(idx[indptr[i]:indptr[i + 1]])) for v in b["lookup_values"][r]],
                                                                 np.float32)) for i in recs)
            assert len(set(b["lookup_values"][r, 1:].tolist())) == 19
            seen.add((int(b["e1"][r]), int(b["rel"][r]), int(b["e2"][r])))
    assert seen == {(int(s["e1"][i]), int(s["rel"][i]), int(t)) for i in range(N) for t in idx[indptr[i]:indptr[i + 1]]}
    # negatives are uniform over the entities: every entity turns up as a negative over a few hundred rows
    counts = np.zeros(E, np.int64)
    for _ in range(10):
        np.add.at(counts, next(it)["lookup_values"].numpy()[:, 1:].reshape(-1), 1)
    assert counts.min() > 0 and counts.max() < 4 * counts.mean()
    with pytest.raises(ValueError):
        DeviceTrainDataset(s, E, 4, num_labels=E + 1, device="cpu")


def test_device_train_dataset_proportional_sampler_rules():
    """The proportional sampler most shipped configs use (one_positive_label_per_sample: False, data.py:228-277) on the device
    path: the leading entries are the record's tails (all of them when there are at most int(L / (1 + prop)) of them, else
    L - min(|E|, L - needed)), in a random order; the rest is the head of a permutation of all entities; labels = membership;
    e2 = the first listed tail."""
    from coper_amd.data import DeviceTrainDataset
    rng = np.random.default_rng(4)
    E, N, L = 97, 40, 22
    indptr, idx = [0], []
    for i in range(N):
        k = int(rng.integers(1, 9))
        idx.extend(sorted(rng.choice(E, size=k, replace=False)))
        indptr.append(len(idx))
    s = dict(e1=np.arange(N), rel=rng.integers(0, 6, N), tail_indptr=np.array(indptr), tail_idx=np.array(idx))
    for prop in (10.0, 2.0):
        need = int(1.0 / (1.0 + prop) * L)
        ds = DeviceTrainDataset(s, E, batch_size=16, num_labels=L, seed=2, device="cpu", one_positive_label_per_sample=False,
                                prop_negatives=prop, shuffle_buffer=30)
        it = iter(ds)
        orders = set()
        for _ in range(12):
            b = {k: v.numpy() for k, v in next(it).items()}
            assert b["lookup_values"].shape == (16, L) and b["e2_multi"].shape == (16, L)
            for r in range(16):
                i = int(b["e1"][r])                                      # e1 = record id in this fixture
                t = idx[indptr[i]:indptr[i + 1]]
                lk, lab = b["lookup_values"][r], b["e2_multi"][r]
                lead = len(t) if len(t) <= need else max(L - min(E, L - need), 0)
                assert set(lk[:lead].tolist()) <= set(t) and len(set(lk[:lead].tolist())) == lead
                assert len(set(lk[lead:].tolist())) == L - lead          # the head of a permutation: distinct
                assert np.array_equal(lab, np.array([float(v in t) for v in lk], np.float32))
                assert int(b["e2"][r]) == int(lk[0]) if lead else int(b["e2"][r]) in t
                assert b["rel"][r] == s["rel"][i]
                if len(t) >= 3 and lead >= 3:
                    orders.add((i, tuple(lk[:3].tolist())))
        if prop == 2.0:
            assert len({i for i, _ in orders}) < len(orders)              # tails come in different orders on different visits


def test_one_vs_all_train_dataset_contract(tmp_path, golden_dir):
    """`train_dataset(..., num_labels=None)` (data.py:157-158 -> `_add_lookup_values`, :314-330): one row per (e1, rel) record,
    dense 0/1 labels over all entities = the record's train tails, empty lookup, e2 = -1 ('None'); host and device-side builders
    of the label matrix agree; the TSV and the synthetic loader both serve it."""
    from coper_amd.data import OneVsAllTrainDataset, SyntheticKGLoader
    from coper_amd.kg_loader import TSVKGLoader
    rng = np.random.default_rng(3)
    E, N = 61, 30
    indptr, idx = [0], []
    for i in range(N):
        idx.extend(sorted(rng.choice(E, size=int(rng.integers(1, 7)), replace=False)))
        indptr.append(len(idx))
    s = dict(e1=rng.integers(0, E, N), rel=rng.integers(0, 4, N), tail_indptr=np.array(indptr), tail_idx=np.array(idx))
    rec_of = {}
    for i in range(N):
        rec_of.setdefault((int(s["e1"][i]), int(s["rel"][i])), []).append(set(idx[indptr[i]:indptr[i + 1]]))
    host = OneVsAllTrainDataset(s, E, batch_size=8, seed=4, shuffle_buffer=10)
    dev = OneVsAllTrainDataset(s, E, batch_size=8, seed=4, shuffle_buffer=10, device="cpu")
    seen = set()
    for k, (bh, bd) in enumerate(zip(host, dev)):
        if k == 12:
            break
        assert bh["e1"].dtype == np.int64 and bh["e2_multi"].dtype == np.float32 and bh["lookup_values"].dtype == np.int32
        assert bh["e2_multi"].shape == (8, E) and bh["lookup_values"].shape == (8, 0) and (bh["e2"] == -1).all()
        for key in ("e1", "rel", "e2", "e2_multi", "lookup_values"):
            assert np.array_equal(bh[key], bd[key].numpy()), key
        for i in range(8):
            tails = set(np.nonzero(bh["e2_multi"][i])[0].tolist())
            assert tails in rec_of[(int(bh["e1"][i]), int(bh["rel"][i]))]
            assert set(np.unique(bh["e2_multi"][i]).tolist()) <= {0.0, 1.0}
            seen.add((int(bh["e1"][i]), int(bh["rel"][i])))
    assert len(seen) >= len(rec_of) - 2                    # the stream repeats and the buffer shuffles: every record comes by
    # the loaders: num_labels=None selects it (config_nations_plain.yaml:22 leaves the key empty)
    for f in ("train.txt", "dev.txt", "test.txt"):
        shutil.copy(os.path.join(golden_dir, "kg_tsv", f), tmp_path)
    ld = TSVKGLoader(str(tmp_path), "nell-995-test")
    ld.assign_ids(write_files=True)
    ld.maybe_create_tf_record_files(str(tmp_path), write_tfrecords=True)
    ts = ld.train_samples()
    tds = ld.train_dataset(None, batch_size=32, num_labels=None, seed=2)
    assert isinstance(tds, OneVsAllTrainDataset)
    tb = next(iter(tds))
    assert tb["e2_multi"].shape == (32, ld.num_ent) and tb["lookup_values"].shape == (32, 0)
    key = {(int(a), int(b)): i for i, (a, b) in enumerate(zip(ts["e1"], ts["rel"]))}
    for i in range(32):
        r = key[(int(tb["e1"][i]), int(tb["rel"][i]))]
        assert np.array_equal(np.nonzero(tb["e2_multi"][i])[0], np.unique(ts["tail_idx"][ts["tail_indptr"][r]:ts["tail_indptr"][r + 1]]))
    from coper_amd.kg_loader import TFRecordKGLoader       # ... and the loader over the reference's own preprocessed directory
    assert isinstance(TFRecordKGLoader(str(tmp_path)).train_dataset(None, batch_size=8, num_labels=None), OneVsAllTrainDataset)
    syn = SyntheticKGLoader("nations_cpg", queries=200)
    ds = syn.train_dataset(None, batch_size=16, num_labels=None)
    assert isinstance(ds, OneVsAllTrainDataset)
    b = next(iter(ds))
    assert b["e2_multi"].shape == (16, syn.num_ent) and b["lookup_values"].shape == (16, 0)
    assert not isinstance(syn.train_dataset(None, batch_size=16, num_labels=10), OneVsAllTrainDataset)
