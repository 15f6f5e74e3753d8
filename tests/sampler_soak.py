#!/usr/bin/env python3
"""Randomised soak of the native samplers (`coper_sample_train_batch`): `python tests/sampler_soak.py [cases] [seed]`.
Per case a random graph (entities 2 ... 2,000,000, records with 1 ... thousands of known tails, sorted or not), a random label count
(1 ... 2,048, up to the number of entities), batch size, sampler and prop_negatives; two batches are drawn and EVERY row is held to the
construction rules of CoPER_ConvE/qa_cpg/data.py:228-311 (the positive / the leading tails, distinct sampled entities in range, labels =
membership in the record's tail list, e1 / rel / e2 of the row's record).  tests/test_gpu_sampler.py holds the distribution tests."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from coper_amd.data import DeviceTrainDataset  # noqa: E402


def _case(rng):
    E = int(rng.choice([2, 3, 7, 40, 211, 3000, 14541, 40943, 250000, 2000000]))
    if rng.random() < 0.3:
        E = int(rng.integers(2, 5000))
    L = int(min(E, rng.choice([1, 2, 3, 17, 64, 100, 257, 1000, 2048])))
    if rng.random() < 0.3:
        L = int(rng.integers(1, min(E, 2048) + 1))
    N = int(rng.integers(1, 60))
    kind = rng.integers(0, 4)
    ks = []
    for _ in range(N):
        if kind == 0:
            k = int(rng.integers(1, 4))
        elif kind == 1:
            k = int(rng.integers(1, min(E, 40) + 1))
        elif kind == 2:
            k = int(rng.integers(1, min(E, 3000) + 1)) if rng.random() < 0.2 else int(rng.integers(1, min(E, 6) + 1))
        else:
            k = int(min(E, rng.choice([1, 2, 9000, 12000]))) if rng.random() < 0.1 else int(rng.integers(1, min(E, 10) + 1))
        ks.append(min(k, E))
    idx = []
    for k in ks:
        t = rng.choice(E, size=k, replace=False) if E < 50 * k else np.unique(rng.integers(0, E, size=k))
        if rng.random() < 0.5:
            t = np.sort(t)
        idx.append(t.astype(np.int64))
    ip = np.zeros(N + 1, np.int64)
    ip[1:] = np.cumsum([len(t) for t in idx])
    s = dict(e1=np.arange(N, dtype=np.int64) % E, rel=rng.integers(0, 9, N), tail_indptr=ip, tail_idx=np.concatenate(idx))
    one_pos = bool(rng.random() < 0.5)
    prop = float(rng.choice([0.0, 1.0, 5.0, 10.0, 100.0]))
    B = int(rng.choice([1, 7, 64, 128]))
    return s, E, L, B, one_pos, prop, idx


def main(argv):
    n_cases = int(argv[0]) if argv else 100
    seed = int(argv[1]) if len(argv) > 1 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    rows = 0
    for c in range(n_cases):
        s, E, L, B, one_pos, prop, tails = _case(rng)
        ds = DeviceTrainDataset(s, E, B, num_labels=L, seed=int(rng.integers(1 << 30)), device="cuda:0", one_positive_label_per_sample=one_pos,
                                prop_negatives=prop)
        assert ds.native, (E, L)
        it = iter(ds)
        need = int(1.0 / (1.0 + prop) * L)
        rec_of = {}                                   # records by e1 (unique while N <= E; otherwise by (e1, rel) candidates)
        for i in range(len(s["e1"])):
            rec_of.setdefault((int(s["e1"][i]), int(s["rel"][i])), []).append(i)
        for _ in range(2):
            b = next(it)
            torch.cuda.synchronize()
            h = {k: v.cpu().numpy() for k, v in b.items()}
            lk, lab = h["lookup_values"], h["e2_multi"]
            assert lk.shape == (B, L) and lab.shape == (B, L) and lk.min() >= 0 and lk.max() < E, (c, E, L)
            for r in range(B):
                cands = rec_of[(int(h["e1"][r]), int(h["rel"][r]))]
                ok = False
                for i in cands:                       # (several records may share (e1, rel) when N > E: one of them explains the row)
                    t = tails[i]
                    ts = set(t.tolist())
                    lead = 1 if one_pos else (len(t) if len(t) <= need else max(L - min(E, L - need), 0))
                    lead = min(lead, L)
                    row = lk[r]
                    good = np.array_equal(lab[r], np.isin(row, t).astype(np.float32))
                    good = good and int(h["e2"][r]) in ts and len(np.unique(row[lead:])) == L - lead
                    if one_pos:
                        good = good and int(row[0]) == int(h["e2"][r])
                    else:
                        good = good and set(row[:lead].tolist()) <= ts and len(np.unique(row[:lead])) == lead and (lead == 0 or int(row[0]) == int(h["e2"][r]))
                    if good:
                        ok = True
                        break
                assert ok, (c, r, E, L, B, one_pos, prop, [len(tails[i]) for i in cands])
                rows += 1
    print("sampler soak: %d cases, %d rows, every row by the construction rules of data.py:228-311; %d s" % (n_cases, rows, time.time() - t0))


if __name__ == "__main__":
    main(sys.argv[1:])
