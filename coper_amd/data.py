"""Synthetic knowledge graphs, the reference batch contract, and sparse (CSR) filters.

The reference loader (`qa_cpg/data.py`) is out of scope as an engine;
what is in scope is the *batch contract* it defines and the evaluator consumes
(`models.py:139-152`, `data.py:168-226`):

    {'e1': int64 [B], 'e2': int64 [B], 'rel': int64 [B],
     'e2_multi': float32 [B, |E|]   (eval: dense 0/1 mask of ALL known tails of (e1, rel)),
     'lookup_values': int32 [B, 0]}  (eval: empty)

This module yields that contract with the mask kept SPARSE (`filt_indptr`, `filt_idx`: the
int64 list the reference stores in its TFRecords before `tf.sparse_to_dense`, data.py:182-186,591)
and a dense-mask adapter for API parity.  No real dataset exists offline
(`/root/reference/.MISSING_LARGE_BLOBS`), so the BASELINE configurations are synthetic KGs of the
named |E|, |R|, d.  NumPy for everything on the host; torch for what lives on the device (`synthetic_entity_rows_device`:
an entity table too large to draw on the host; `DeviceTrainDataset`: the samplers on the GPU).  Nothing here touches the oracle."""
from __future__ import annotations

import math
from typing import Dict, Iterator, Optional

import numpy as np
import torch

__all__ = ["CONFIGS", "model_descriptors", "synthetic_params", "reference_init_params", "synthetic_queries", "SyntheticKGLoader",
           "synthetic_entity_rows_device", "ENTITY_SEED_BLOCK",
           "dense_filter_to_csr", "csr_to_dense_filter", "param_shapes"]

_COMMON = dict(use_negative_sampling=False, label_smoothing_epsilon=0.1, input_dropout=0.2, hidden_dropout=0.3,
               output_dropout=0.2, add_loss_summaries=False, add_variable_summaries=False,
               add_tensor_summaries=False, learning_rate=0.001, concat_rel=False, context_rel_dropout=0.2,
               context_rel_use_batch_norm=True, batch_norm_momentum=0.99, batch_norm_train_stats=True)

# BASELINE.json configs, made concrete as in BASELINE.md section 2 (R2 = 2|R|: `_reverse` relations
# get their own ids, data.py:422-428).  `queries` = queries per evaluation pass.
CONFIGS = {
    # CoPER-ConvE on Nations-shaped data, plumbing (config_nations_cpg.yaml shapes with BASELINE's d=32)
    "nations_cpg": dict(num_ent=14, num_rel=110, ent_emb_size=32, rel_emb_size=8, emb_h=4, emb_w=8,
                        context_rel_conv=None, context_rel_out=[], queries=2000),
    # config_FB15k-237_cpg.yaml
    "fb15k237_cpg": dict(num_ent=14541, num_rel=474, ent_emb_size=200, rel_emb_size=32,
                         context_rel_conv=None, context_rel_out=[], queries=20480),
    # config_WN18RR_cpg.yaml
    "wn18rr_cpg": dict(num_ent=40943, num_rel=22, ent_emb_size=200, rel_emb_size=8,
                       context_rel_conv=None, context_rel_out=[], queries=3072),
    # config_FB15k-237_plain.yaml
    "fb15k237_plain": dict(num_ent=14541, num_rel=474, ent_emb_size=200, rel_emb_size=200,
                           context_rel_conv=None, context_rel_out=None, queries=20480),
    # synthetic 10M-entity KG (BASELINE.json configs[4])
    "synth10m_cpg": dict(num_ent=10_000_000, num_rel=2000, ent_emb_size=256, rel_emb_size=32, emb_h=16, emb_w=16,
                         context_rel_conv=None, context_rel_out=[], queries=4096),
}


def model_descriptors(name: str, **overrides) -> dict:
    """The `model_descriptors` dict `run_cpg.py:115-137` builds, for a named BASELINE config."""
    md = dict(_COMMON)
    md.update(CONFIGS[name])
    md.pop("queries", None)
    md.update(overrides)
    return md


def _dims(md):
    d, r = int(md["ent_emb_size"]), int(md["rel_emb_size"])
    fh, fw, C = int(md.get("conv_filter_height", 3)), int(md.get("conv_filter_width", 3)), int(md.get("conv_num_channels", 32))
    emb_h = int(md.get("emb_h", 10))
    emb_w = int(md.get("emb_w", d // emb_h))
    lookup = bool(md.get("do_parameter_lookup", False))
    gen_conv = md.get("context_rel_conv", None) is not None
    gen_fc = md.get("context_rel_out", None) is not None
    stacked = (not gen_conv) and (not gen_fc) and (not lookup)
    in_h = 2 * emb_h if stacked else emb_h
    Ho, Wo = in_h - fh + 1, emb_w - fw + 1
    F = Ho * Wo * C + (r if md.get("concat_rel", False) else 0)
    return dict(d=d, r=r, fh=fh, fw=fw, C=C, emb_h=emb_h, emb_w=emb_w, lookup=lookup, gen_conv=gen_conv,
                gen_fc=gen_fc, stacked=stacked, Ho=Ho, Wo=Wo, F=F)


def param_shapes(md: dict) -> Dict[str, tuple]:
    """Leaf name -> shape of every variable `ConvE._create_variables` creates (models.py:203-336),
    plus the BN variables of `tf.layers.batch_normalization` (models.py:63-65,386-388,416-418)."""
    dm = _dims(md)
    E, R, d, r, C = int(md["num_ent"]), int(md["num_rel"]), dm["d"], dm["r"], dm["C"]
    F = dm["F"]
    sh = {"ent_emb": (E, d), "pred_bias": (E,)}
    if not dm["lookup"]:
        sh["rel_emb"] = (R, r)
    ctx_bn = bool(md.get("context_rel_use_batch_norm", False))

    def gen(name, hidden, n_out):
        size_in = r
        for i, n in enumerate(list(hidden) + [n_out]):
            sh["%s/CPG/Projection%d" % (name, i)] = (size_in, n)
            if i < len(hidden) and ctx_bn:
                for leaf in ("gamma", "beta", "moving_mean", "moving_variance"):
                    sh["%s/CPG/Projection%d/BatchNorm/%s" % (name, i, leaf)] = (n,)
            size_in = n

    nconv = dm["fh"] * dm["fw"] * C
    if dm["gen_conv"]:
        if dm["lookup"]:
            sh["conv1_weights"] = (R, nconv)
            sh["conv1_bias"] = (R, C)
        else:
            gen("conv1_weights", md["context_rel_conv"], nconv)
            gen("conv1_bias", md["context_rel_conv"], C)
    else:
        sh["conv1_weights"] = (dm["fh"], dm["fw"], 1, C)
        sh["conv1_bias"] = (C,)
    if dm["gen_fc"]:
        if dm["lookup"]:
            sh["fc_weights"] = (R, F * d)
            sh["fc_bias"] = (R, d)
        else:
            gen("fc_weights", md["context_rel_out"], F * d)
            gen("fc_bias", md["context_rel_out"], d)
    else:
        sh["fc_weights"] = (F, d)
        sh["fc_bias"] = (d,)
    for leaf in ("gamma", "beta", "moving_mean", "moving_variance"):
        sh["Conv1BN/" + leaf] = (C,)
        sh["FCBN/" + leaf] = (d,)
    return sh


def synthetic_params(md: dict, seed: int = 0, skip=(), ent_std: float = 0.1) -> Dict[str, np.ndarray]:
    """Random-init weights of the named architecture with O(1) activations by construction
    (SURVEY 8(d)): entity rows ~ N(0, 0.1^2) (rounds 1 - 3 drew 0.3: the survey's law since round 4); filters / dense weights scaled so that the
    pre-BN activations have ~unit variance; BN statistics close to the analytic moments, with
    gamma ~ U(0.5,1.5), beta ~ N(0,0.1^2); pred_bias ~ N(0,0.1^2).  Logits come out O(1-10),
    which makes the 1e-3 parity gate meaningful.  `skip`: leaf names not to materialise here
    (e.g. 'ent_emb' of the 10M-entity config, generated on the device instead)."""
    rng = np.random.default_rng(seed)
    dm = _dims(md)
    shapes = param_shapes(md)
    d, r, C, F = dm["d"], dm["r"], dm["C"], dm["F"]
    s_e, s_c = float(ent_std), 0.3      # (ent_std = 0.3: the table of rounds 1 - 3, kept by the training tests whose bounds were set on it)
    p = {}

    def normal(shape, std):
        return (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

    for name, shape in shapes.items():
        if name in skip:
            continue
        leaf = name.rsplit("/", 1)[-1]
        if name == "ent_emb":
            p[name] = normal(shape, s_e)
        elif name == "rel_emb":
            p[name] = normal(shape, s_e if dm["stacked"] else s_c)
        elif name == "pred_bias":
            p[name] = normal(shape, 0.1)
        elif leaf == "gamma":
            p[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf == "beta":
            p[name] = normal(shape, 0.1)
        elif leaf == "moving_mean":
            p[name] = normal(shape, 0.1)
        elif leaf == "moving_variance":
            p[name] = rng.uniform(0.8, 1.25, shape).astype(np.float32)
        elif "/CPG/Projection" in name:
            base = name.split("/", 1)[0]
            fan_in = shape[0]
            idx = int(name.rsplit("Projection", 1)[1])
            hidden = md["context_rel_conv"] if base.startswith("conv1") else md["context_rel_out"]
            last = idx == len(hidden)
            if last:
                ctx_std = s_c if idx == 0 else 0.6  # relu(BN(.)) hidden activations ~ 0.6 rms
                target = {"conv1_weights": 1.0 / (s_e * 3.0), "conv1_bias": 0.1,
                          "fc_weights": 1.0 / math.sqrt(0.5 * F), "fc_bias": 0.1}[base]
                p[name] = normal(shape, target / (ctx_std * math.sqrt(fan_in)))
            else:
                ctx_std = s_c if idx == 0 else 0.6
                p[name] = normal(shape, 1.0 / (ctx_std * math.sqrt(fan_in)))
        elif name == "conv1_weights":
            p[name] = normal(shape, 1.0 / (s_e * 3.0))
        elif name == "conv1_bias":
            p[name] = normal(shape, 0.1)
        elif name == "fc_weights":
            p[name] = normal(shape, 1.0 / math.sqrt(0.5 * F))
        elif name == "fc_bias":
            p[name] = normal(shape, 0.1)
        else:  # pragma: no cover
            raise KeyError(name)
    return p


def reference_init_params(md: dict, seed: int = 0, skip=()) -> Dict[str, np.ndarray]:
    """The variables as `ConvE._create_variables` INITIALISES them (models.py:203-336): `xavier_initializer()` -- uniform,
    limit sqrt(6 / (fan_in + fan_out)) -- for ent_emb, rel_emb, the static conv / dense weights and the generator projections
    of the weights (models.py:205-214,238,253-256,291,305-308); zeros for conv1_bias, fc_bias, their generators' projections
    and pred_bias (models.py:246,258,301,310,312-314); BN at its `tf.layers.batch_normalization` defaults (gamma 1, beta 0,
    moving mean 0, moving variance 1).  An untrained reference model: entity elements within +-0.0202 at FB15k-237's shape,
    +-7.7e-4 for a 10M-entity table, embeddings h of ~1e-3 -- the operand scales the bf16x3 arithmetic has to carry without
    help from the data (tests/test_gpu_scale.py).  g_lookup tables take the projections' law."""
    rng = np.random.default_rng(seed)
    dm = _dims(md)
    p = {}

    def xavier(shape):
        if len(shape) == 4:      # conv filter [fh, fw, in, out]: receptive field x channels
            rf = shape[0] * shape[1]
            fan_in, fan_out = rf * shape[2], rf * shape[3]
        else:
            fan_in, fan_out = shape[0], shape[-1]
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, shape).astype(np.float32)

    for name, shape in param_shapes(md).items():
        if name in skip:
            continue
        leaf = name.rsplit("/", 1)[-1]
        base = name.split("/", 1)[0]
        if leaf in ("gamma", "moving_variance"):
            p[name] = np.ones(shape, np.float32)
        elif leaf in ("beta", "moving_mean") or name == "pred_bias":
            p[name] = np.zeros(shape, np.float32)
        elif base in ("conv1_bias", "fc_bias"):
            p[name] = np.zeros(shape, np.float32)
        else:
            p[name] = xavier(shape)
    return p


ENTITY_SEED_BLOCK = 1 << 16   # rows per independently seeded block of a device-drawn entity table


def synthetic_entity_rows_device(md: dict, seed: int, device, lo: int = 0, hi: Optional[int] = None):
    """Rows [lo, hi) of the synthetic entity table of a config too large for the host generator
    (`ent_emb ~ N(0, 0.1^2)`, `pred_bias ~ N(0, 0.1^2)`, the laws of `synthetic_params`), drawn on `device`.
    The table is cut into blocks of ENTITY_SEED_BLOCK rows and block b is drawn from its own generator seeded
    with (seed, b), so a row's values do not depend on which shard, or how many shards, hold it: every
    sharding of the table scores the same KG (entity-sharded ranks can be compared across world sizes).
    Returns (ent_emb [hi-lo, d] f32, pred_bias [hi-lo] f32)."""
    import torch
    E, d = int(md["num_ent"]), int(md["ent_emb_size"])
    hi = E if hi is None else int(hi)
    lo = int(lo)
    ent = torch.empty((hi - lo, d), device=device, dtype=torch.float32)
    bias = torch.empty((hi - lo,), device=device, dtype=torch.float32)
    g = torch.Generator(device=device)
    for b in range(lo // ENTITY_SEED_BLOCK, (hi + ENTITY_SEED_BLOCK - 1) // ENTITY_SEED_BLOCK):
        b0 = b * ENTITY_SEED_BLOCK
        n = min(ENTITY_SEED_BLOCK, E - b0)
        g.manual_seed((int(seed) * 1000003 + b) * 2 + 1)
        rows = torch.randn((n, d), generator=g, device=device, dtype=torch.float32) * 0.1
        bs = torch.randn((n,), generator=g, device=device, dtype=torch.float32) * 0.1
        s0, s1 = max(lo, b0), min(hi, b0 + n)
        ent[s0 - lo:s1 - lo] = rows[s0 - b0:s1 - b0]
        bias[s0 - lo:s1 - lo] = bs[s0 - b0:s1 - b0]
    return ent, bias


def synthetic_queries(md: dict, Q: int, seed: int = 0, mean_filter: float = 4.0, max_filter: int = 64,
                      order: str = "shuffled"):
    """Evaluation queries + CSR known-answer filters: e1, e2 ~ U[0,|E|); rel ~ U[0,|R|)
    over the FORWARD half of the R2 table (eval excludes inverse relations, run_cpg.py:156,164,172);
    filter list = {e2} U Geometric(mean 4, cap 64) uniform entities, sorted unique.
    order: 'shuffled' | 'sorted' (by relation -- exposes the per-relation weight reuse)."""
    rng = np.random.default_rng(seed + 1000003)
    E, R2 = int(md["num_ent"]), int(md["num_rel"])
    Rf = max(1, R2 // 2)
    e1 = rng.integers(0, E, Q, dtype=np.int64)
    rel = rng.integers(0, Rf, Q, dtype=np.int64)
    e2 = rng.integers(0, E, Q, dtype=np.int64)
    if order == "sorted":
        o = np.argsort(rel, kind="stable")
        e1, rel, e2 = e1[o], rel[o], e2[o]
    n_extra = np.minimum(rng.geometric(1.0 / (1.0 + mean_filter), Q) - 1, max_filter - 1)
    n_extra = np.minimum(n_extra, E - 1)
    indptr = np.zeros(Q + 1, dtype=np.int64)
    rows = []
    for i in range(Q):
        extra = rng.integers(0, E, int(n_extra[i]), dtype=np.int64)
        row = np.unique(np.concatenate([extra, e2[i:i + 1]]))
        rows.append(row)
        indptr[i + 1] = indptr[i] + len(row)
    idx = np.concatenate(rows) if rows else np.zeros(0, np.int64)
    return dict(e1=e1, rel=rel, e2=e2, filt_indptr=indptr, filt_idx=idx.astype(np.int64))


def dense_filter_to_csr(e2_multi, device=None):
    """Dense 0/1 mask [B,|E|] (data.py:182-186) -> (indptr int64 [B+1], idx int64 sorted).
    With a HIP `device` the scan runs there (the mask crosses PCIe once: 0.7 ms per 512 x 14,541 batch on the MI355X box
    against 8.6 ms for the host scan -- the reference's batch contract carries these masks, 30 MB per batch)."""
    if device is not None and torch.cuda.is_available():
        m = torch.as_tensor(np.ascontiguousarray(e2_multi)).to(device) == 1
        counts = m.sum(dim=1, dtype=torch.int64)
        cols = m.nonzero()[:, 1]                       # row-major: ascending inside a row
        indptr = torch.zeros(m.shape[0] + 1, dtype=torch.int64, device=m.device)
        indptr[1:] = torch.cumsum(counts, dim=0)
        return indptr.cpu().numpy(), cols.cpu().numpy().astype(np.int64)
    rows, cols = np.nonzero(np.asarray(e2_multi) == 1)
    B = e2_multi.shape[0]
    indptr = np.zeros(B + 1, dtype=np.int64)
    np.add.at(indptr, rows + 1, 1)
    return np.cumsum(indptr), cols.astype(np.int64)


def canonical_csr(indptr, idx):
    """The rank kernels take every row of the CSR filter sorted ascending (include/coper_hip.h; repeated ids are harmless
    when adjacent -- the dense mask of metrics.py:45 is idempotent).  Rows that come unsorted (hand-built lists) are sorted
    here; rows already in order -- everything this package's loaders produce -- cost one vectorised check."""
    indptr = np.asarray(indptr, np.int64)
    idx = np.asarray(idx, np.int64)
    if len(idx) < 2:
        return indptr, idx
    down = idx[1:] < idx[:-1]                   # (one pass, no int64 temporary: this check is most of a pass's host marshalling)
    inner = indptr[1:-1]
    if len(inner) and (inner[0] <= 0 or inner[-1] >= len(idx)):      # (empty rows at either end; indptr is non-decreasing)
        inner = inner[np.searchsorted(inner, 0, side="right"):np.searchsorted(inner, len(idx), side="left")]
    inner = np.unique(inner) if len(inner) > 1 and np.any(inner[1:] == inner[:-1]) else inner      # (empty rows repeat a boundary)
    # steps down across row boundaries do not count: in order iff every step down sits on a boundary
    if np.count_nonzero(down) == np.count_nonzero(down[inner - 1]):
        return indptr, idx
    down[inner - 1] = False
    out = idx.copy()
    for i in np.unique(np.searchsorted(indptr, np.nonzero(down)[0], side="right") - 1):
        out[indptr[i]:indptr[i + 1]] = np.sort(idx[indptr[i]:indptr[i + 1]], kind="stable")
    return indptr, out


def csr_to_dense_filter(indptr, idx, num_ent: int) -> np.ndarray:
    B = len(indptr) - 1
    m = np.zeros((B, num_ent), dtype=np.float32)
    for i in range(B):
        m[i, idx[indptr[i]:indptr[i + 1]]] = 1.0
    return m


class SyntheticKGLoader(object):
    """Mirror of the reference loader surface the driver touches (data.py:25-29,82-100,168-174):
    `dataset_name`, `num_ent`, `num_rel`, `needs_test_set_cleaning`, `eval_dataset(...)`.
    `eval_dataset` returns a re-iterable of batches in the reference batch contract with the filter
    in CSR form (`filt_indptr`, `filt_idx`); `dense_mask=True` also materialises `e2_multi`."""

    def __init__(self, config_name: str = "fb15k237_cpg", seed: int = 0, queries: Optional[int] = None,
                 md: Optional[dict] = None):
        self.dataset_name = config_name
        self.md = md if md is not None else model_descriptors(config_name)
        self.num_ent = int(self.md["num_ent"])
        self.num_rel = int(self.md["num_rel"])
        self.needs_test_set_cleaning = False
        self.seed = seed
        self.queries = queries if queries is not None else CONFIGS.get(config_name, {}).get("queries", 1024)
        self._cache = {}

    def maybe_create_tf_record_files(self, directory=None, buffer_size=None):
        return None  # nothing to create: the KG is synthetic

    def _queries(self, dataset_type):
        if dataset_type not in self._cache:
            salt = {"train": 1, "dev": 2, "test": 3}.get(dataset_type, 4)
            self._cache[dataset_type] = synthetic_queries(self.md, self.queries, seed=self.seed * 7 + salt)
        return self._cache[dataset_type]

    def eval_dataset(self, directory=None, dataset_type="test", batch_size=512, include_inv_relations=False,
                     buffer_size=None, prefetch_buffer_size=None, dense_mask=False):
        q = self._queries(dataset_type)
        return EvalDataset(q, batch_size, self.num_ent, dense_mask)

    def train_samples(self):
        """The synthetic train graph as the records a train TFRecord holds (data.py:481-489, 574-594): one per distinct (e1, rel)
        of the synthetic "train" queries, with the union of their known-answer lists as its tails."""
        q = self._queries("train")
        key = q["e1"].astype(np.int64) * self.num_rel + q["rel"]
        order = np.argsort(key, kind="stable")
        uniq, first = np.unique(key[order], return_index=True)
        bounds = list(first) + [len(order)]
        e1, rel, indptr, idx = [], [], [0], []
        for u in range(len(uniq)):
            rows = order[bounds[u]:bounds[u + 1]]
            tails = np.unique(np.concatenate([q["filt_idx"][q["filt_indptr"][i]:q["filt_indptr"][i + 1]] for i in rows] + [q["e2"][rows]]))
            e1.append(int(q["e1"][rows[0]])); rel.append(int(q["rel"][rows[0]]))
            idx.extend(int(t) for t in tails)
            indptr.append(len(idx))
        return dict(e1=np.asarray(e1, np.int64), rel=np.asarray(rel, np.int64), tail_indptr=np.asarray(indptr, np.int64),
                    tail_idx=np.asarray(idx, np.int64))

    def train_dataset(self, directory=None, batch_size=512, include_inv_relations=True, num_parallel_readers=None,
                      num_parallel_batches=None, buffer_size=None, prefetch_buffer_size=None, prop_negatives=10.0,
                      num_labels=100, cache=False, one_positive_label_per_sample=True, seed=0, device=None):
        """The reference loader's `train_dataset` signature (data.py:89-100) over the synthetic graph; `num_labels=None` = 1-vs-all
        labels (data.py:157-158); `device` (extra keyword): sample / densify on that device."""
        samples = self.train_samples()
        if num_labels is None:
            return OneVsAllTrainDataset(samples, self.num_ent, batch_size, seed, device=device)
        if device is not None:
            return DeviceTrainDataset(samples, self.num_ent, batch_size, num_labels, seed, device=device,
                                      one_positive_label_per_sample=one_positive_label_per_sample, prop_negatives=prop_negatives)
        return TrainDataset(samples, self.num_ent, batch_size, num_labels, one_positive_label_per_sample, prop_negatives, seed)


class EvalDataset(object):
    """Re-iterable batch source; exhausting it is the end-of-data signal the reference gets as
    `tf.errors.OutOfRangeError` (metrics.py:59)."""

    def __init__(self, q, batch_size, num_ent, dense_mask=False):
        self.q, self.batch_size, self.num_ent, self.dense_mask = q, int(batch_size), num_ent, dense_mask

    def __len__(self):
        return (len(self.q["e1"]) + self.batch_size - 1) // self.batch_size

    @property
    def num_queries(self):
        return len(self.q["e1"])

    def as_single_batch(self):
        q = self.q
        return dict(e1=q["e1"], e2=q["e2"], rel=q["rel"], filt_indptr=q["filt_indptr"], filt_idx=q["filt_idx"],
                    lookup_values=np.zeros((len(q["e1"]), 0), np.int32))

    def staged_for(self, model):
        """The whole dataset marshalled once for `model` (ConvE.stage_persistent: canonical CSR, int32 in pinned memory, device
        buffers): `ranking_and_hits` scores the same evaluation set after every epoch (run_cpg.py:18-35, 228-250) and pays
        the host-side marshalling once.  One entry per model, dropped with it; None when the model has no such path."""
        out = self.staged_chunks_for(model, None)
        return out[0] if out else None

    def staged_chunks_for(self, model, max_chunk=None):
        """`staged_for` for sets of any size: the dataset cut into ceil(Q / max_chunk) chunks of (nearly) equal size, each marshalled
        once (its own pinned buffer, device arrays and rank buffer) -- FB15k-237's test set is 40,932 queries with both directions,
        more than one device pass takes.  The list (one entry when max_chunk is None or the set fits); None when the model has no
        such path or an id does not fit int32."""
        if not hasattr(model, "stage_persistent"):
            return None
        import weakref
        cache = self.__dict__.setdefault("_staged", {})
        key = (id(model), None if max_chunk is None else int(max_chunk))
        ent = cache.get(key)
        if ent is not None and ent[0]() is model:
            return ent[1]
        q = self.q
        Q = len(q["e1"])
        ip, ix = canonical_csr(q["filt_indptr"], q["filt_idx"])
        ip = np.asarray(ip)
        n_chunks = 1 if (max_chunk is None or Q <= max_chunk) else -(-Q // int(max_chunk))
        size = -(-Q // n_chunks) if Q else 0
        sbs = []
        for c in range(n_chunks):
            lo, hi = c * size, min(Q, (c + 1) * size)
            cip = ip[lo:hi + 1]
            sb = model.stage_persistent(q["e1"][lo:hi], q["rel"][lo:hi], q["e2"][lo:hi], cip - cip[0], ix[cip[0]:cip[-1]])
            if sb is None:
                return None
            sbs.append(sb)
        # the entries (pinned host buffers and int64 device arrays) go when the model does: the callback drops them, unless the
        # slot was taken over by a later model that got the same id()
        def _drop(ref, cache=cache, key=key):
            if cache.get(key, (None,))[0] is ref:
                del cache[key]
        cache[key] = (weakref.ref(model, _drop), sbs)
        return sbs

    def __iter__(self) -> Iterator[dict]:
        q = self.q
        Q = len(q["e1"])
        for s in range(0, Q, self.batch_size):
            e = min(Q, s + self.batch_size)
            ip = q["filt_indptr"][s:e + 1]
            batch = dict(e1=q["e1"][s:e], e2=q["e2"][s:e], rel=q["rel"][s:e],
                         filt_indptr=(ip - ip[0]).astype(np.int64), filt_idx=q["filt_idx"][ip[0]:ip[-1]],
                         lookup_values=np.zeros((e - s, 0), np.int32))
            if self.dense_mask:
                batch["e2_multi"] = csr_to_dense_filter(batch["filt_indptr"], batch["filt_idx"], self.num_ent)
            yield batch


class DeviceTrainDataset(object):
    """The samplers of `TrainDataset` -- one positive per row (data.py:138-144, 278-311) and the proportional one most shipped
    configs use (`one_positive_label_per_sample: False`, data.py:228-277; `_device_batch_prop`) -- with the heavy parts on the device: the
    host sampler takes ~40 ms per 512 x 1000 batch on one core (the reference spreads it over 32 `tf.data` map threads,
    data.py:93-94), a training step 1.1 ms.

    Host (cheap, exact): the record stream, its expansion into one row per known tail, the shuffle buffer of 1000 rows and
    the batching -- on row ids only.  Device: for every row a window of `num_labels - 1` consecutive entries of a fresh
    uniform permutation of the entities (= an ordered uniform sample without replacement: the keys of `torch.rand` sorted),
    the positive in front, and the labels = membership of the looked-up ids in the row's tail list (so a sampled
    "negative" that is a known tail is supervised as positive, as the reference comments).  One deviation, distributional
    like the rest (TF's RNG stream cannot be reproduced): every ROW draws its own permutation; the reference draws one per
    record and gives its rows different windows of it.
    Batches are dicts of device tensors in the reference's batch contract (models.py:139-152); `ConvE.train_step` takes
    them as they are."""

    def __init__(self, samples, num_ent, batch_size, num_labels=100, seed=0, shuffle_buffer=1000, device="cuda:0",
                 one_positive_label_per_sample=True, prop_negatives=10.0, native=None):
        """native: build the batch with the library's sampler kernel (`coper_sample_train_batch`: one launch, nothing read back;
        round 6) -- the default on a GPU when num_labels <= 2048; False: the torch-op construction of rounds 2 - 5 (the only one on
        the CPU; the same distribution, another random stream)."""
        self.num_ent, self.batch_size, self.num_labels = int(num_ent), int(batch_size), int(num_labels)
        self.one_pos, self.prop = bool(one_positive_label_per_sample), float(prop_negatives)
        if self.num_labels > self.num_ent:
            raise ValueError("num_labels needs to be at most the total number of entities (data.py:146-147)")
        self.shuffle_buffer = int(shuffle_buffer)
        self.device = torch.device(device)
        self.rng = np.random.default_rng(seed)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed) + 12345)
        ip = np.asarray(samples["tail_indptr"], np.int64)
        k = np.diff(ip)
        self.n_rec = len(k)
        # rows in record order: (record, position of the positive inside the record's tail list)
        self.row_rec = np.repeat(np.arange(self.n_rec, dtype=np.int64), k)
        self.row_tail = np.asarray(samples["tail_idx"], np.int64)          # the positive of each row (same order)
        self.rec_first_row = ip[:-1]
        self.n_rows = len(self.row_rec)
        if self.n_rows == 0:
            raise ValueError("DeviceTrainDataset: no (e1, rel) record has a known tail -- nothing to sample from")
        if not self.one_pos:      # the proportional sampler (data.py:228-277): one row per record that has tails
            self.rows_of_rec = np.nonzero(k > 0)[0].astype(np.int64)
        self.d_e1 = torch.as_tensor(np.asarray(samples["e1"], np.int64)).to(self.device)
        self.d_rel = torch.as_tensor(np.asarray(samples["rel"], np.int64)).to(self.device)
        self.d_ip = torch.as_tensor(ip).to(self.device)
        self.d_tails = torch.as_tensor(self.row_tail).to(self.device)
        self._k = k
        can_native = self.device.type == "cuda" and self.num_labels <= 2048 and self.num_ent < 2 ** 31
        if native and not can_native:
            raise ValueError("DeviceTrainDataset(native=True) needs a GPU, num_labels <= 2048 and num_ent < 2^31")
        self.native = can_native if native is None else bool(native)
        self._seed, self._batch_no = int(seed) & (2 ** 64 - 1), 0
        if self.native:
            from . import _lib
            self._lib = _lib.load()
            # the batch's record ids (and positives) travel through a ring of pinned buffers: one asynchronous copy per batch
            self._pins = [torch.empty(2 * self.batch_size, dtype=torch.int64).pin_memory() for _ in range(4)]
            self._pin_ev = [None] * 4

    def _negatives(self, B, n=None):
        """[B, n] int64 (n = L - 1 by default): per row the first n entries of a fresh uniform permutation of the entities = an ordered uniform
        sample without replacement.  Few labels against many entities: uniform draws in order, repeats skipped (the same
        distribution; sorts M = L - 1 + slack draws per row instead of |E| keys); otherwise, or if a row runs short of distinct draws, the sorted-keys form."""
        dev, E = self.device, self.num_ent
        n = self.num_labels - 1 if n is None else int(n)
        if n <= 0:
            return torch.zeros((B, 0), dtype=torch.int64, device=dev)
        if 4 * n <= E:
            M = n + max(32, int(2.0 * n * n / E) + 8 * int(math.sqrt(max(n * n / (2.0 * E), 1.0))))   # expected repeats n^2 / 2E
            draws = torch.randint(0, E, (B, M), device=dev, generator=self.gen)
            vals, order = torch.sort(draws, dim=1, stable=True)
            dup_sorted = torch.zeros((B, M), dtype=torch.bool, device=dev)
            dup_sorted[:, 1:] = vals[:, 1:] == vals[:, :-1]          # stable sort: the later draw of a repeat comes second
            dup = torch.zeros((B, M), dtype=torch.bool, device=dev)
            dup.scatter_(1, order, dup_sorted)
            keep = ~dup
            rank = torch.cumsum(keep, dim=1)
            if bool((rank[:, -1] >= n).all()):
                sel = keep & (rank <= n)
                return draws[sel].view(B, n)
        keys = torch.rand((B, E), device=dev, generator=self.gen)
        return torch.argsort(keys, dim=1)[:, :n]

    def _native_batch(self, ids):
        """One launch of `coper_sample_train_batch` on the current stream.  ids: rows (one positive per row) or records (proportional)."""
        dev, L, B = self.device, self.num_labels, len(ids)
        slot = self._batch_no % len(self._pins)
        if self._pin_ev[slot] is not None:
            self._pin_ev[slot].synchronize()          # (the copy that read this buffer four batches ago: long done)
        pin = self._pins[slot]
        host = pin.numpy()
        if self.one_pos:
            rec = self.row_rec[ids]
            host[:B] = rec
            host[B:2 * B] = self.row_tail[ids]
        else:
            rec = np.asarray(ids, np.int64)
            host[:B] = rec
        staged = torch.empty(2 * B, dtype=torch.int64, device=dev)
        staged.copy_(pin[:2 * B], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._pin_ev[slot] = ev
        out = dict(e1=torch.empty(B, dtype=torch.int64, device=dev), rel=torch.empty(B, dtype=torch.int64, device=dev),
                   e2=torch.empty(B, dtype=torch.int64, device=dev), lookup_values=torch.empty((B, L), dtype=torch.int32, device=dev),
                   e2_multi=torch.empty((B, L), dtype=torch.float32, device=dev))
        rc = self._lib.coper_sample_train_batch(
            dev.index or 0, staged.data_ptr(), staged.data_ptr() + 8 * B if self.one_pos else None, self.d_e1.data_ptr(), self.d_rel.data_ptr(),
            self.d_ip.data_ptr(), self.d_tails.data_ptr(), B, L, self.num_ent, 0 if self.one_pos else 1, self.prop, int(self._k[rec].max()),
            self._seed, self._batch_no, out["e1"].data_ptr(), out["rel"].data_ptr(), out["e2"].data_ptr(), out["lookup_values"].data_ptr(),
            out["e2_multi"].data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:
            raise RuntimeError("coper_sample_train_batch failed with status %d" % rc)
        # (`staged` is freed with this frame: the caching allocator hands its memory out again in THIS stream's order, behind the launch)
        self._batch_no += 1
        return out

    def _device_batch(self, rows):
        if self.native:
            return self._native_batch(rows)
        dev, E, L, B = self.device, self.num_ent, self.num_labels, len(rows)
        rows_d = torch.as_tensor(rows).to(dev)
        rec = torch.as_tensor(self.row_rec[rows]).to(dev)
        e2 = self.d_tails[rows_d]
        neg = self._negatives(B)
        lookup = torch.cat([e2[:, None], neg], dim=1)
        # labels: membership in the row's tail list
        lo, hi = self.d_ip[rec], self.d_ip[rec + 1]
        cnt = hi - lo
        owner = torch.repeat_interleave(torch.arange(B, device=dev), cnt)
        pos = torch.arange(int(cnt.sum().item()), device=dev) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt)
        member = torch.zeros((B, E), dtype=torch.float32, device=dev)
        member[owner, self.d_tails[lo[owner] + pos]] = 1.0
        labels = torch.gather(member, 1, lookup)
        return dict(e1=self.d_e1[rec], rel=self.d_rel[rec], e2=e2, lookup_values=lookup.to(torch.int32), e2_multi=labels)

    def _device_batch_prop(self, recs):
        """data.py:228-277: per record the tails in a fresh random order, then the head of a fresh permutation of ALL entities;
        num_positives_needed = int(L / (1 + prop_negatives)); a record with more tails than that keeps L - min(|E|, L - needed)
        of them.  Labels = membership in the tail list; e2 = the first tail of the shuffled order."""
        if self.native:
            return self._native_batch(recs)
        dev, E, L, B = self.device, self.num_ent, self.num_labels, len(recs)
        rec = torch.as_tensor(recs).to(dev)
        lo, hi = self.d_ip[rec], self.d_ip[rec + 1]
        cnt = hi - lo
        tot = int(cnt.sum().item())
        owner = torch.repeat_interleave(torch.arange(B, device=dev), cnt)
        first = torch.cumsum(cnt, 0) - cnt
        pos = torch.arange(tot, device=dev) - first[owner]
        tails = self.d_tails[lo[owner] + pos]
        # random order inside every record: sort by (owner, uniform key)
        order = torch.argsort(owner.to(torch.float64) + torch.rand(tot, device=dev, generator=self.gen, dtype=torch.float64))
        tails_sh = tails[order]                      # segments stay contiguous (owner is the integer part), shuffled inside
        need = int(1.0 / (1.0 + self.prop) * L)
        n_neg_big = min(E, L - need)
        lead = torch.where(cnt <= need, cnt, torch.full_like(cnt, L - n_neg_big)).clamp(max=L)      # tails kept per row
        neg = self._negatives(B, min(L, E))
        grid = torch.arange(L, device=dev)[None, :]
        idx = torch.gather(neg, 1, (grid - lead[:, None]).clamp(min=0, max=neg.shape[1] - 1))
        keep = pos < lead[owner]                     # (pos = rank inside the shuffled segment as well)
        idx[owner[keep], pos[keep]] = tails_sh[keep]
        member = torch.zeros((B, E), dtype=torch.float32, device=dev)
        member[owner, tails] = 1.0
        labels = torch.gather(member, 1, idx)
        e2 = tails_sh[first]                         # every record here has at least one tail
        return dict(e1=self.d_e1[rec], rel=self.d_rel[rec], e2=e2, lookup_values=idx.to(torch.int32), e2_multi=labels)

    def _row_batches(self):
        buf, pos = [], 0        # shuffle buffer of row ids; the record stream repeats (.repeat())
        while True:
            need = self.shuffle_buffer + self.batch_size - len(buf)
            while need > 0:
                if not self.one_pos:                 # rows are records
                    buf.append(int(self.rows_of_rec[pos % len(self.rows_of_rec)]))
                    pos += 1
                    need -= 1
                    continue
                i = pos % self.n_rec
                pos += 1
                first = int(self.rec_first_row[i])
                last = int(self.rec_first_row[i + 1]) if i + 1 < self.n_rec else self.n_rows
                buf.extend(range(first, last))
                need -= last - first
            js = self.rng.integers(0, 1 << 62, size=self.batch_size)
            take = []
            for t in range(self.batch_size):
                j = int(js[t] % min(len(buf), self.shuffle_buffer))
                take.append(buf[j])
                buf[j] = buf[-1]
                buf.pop()
            yield np.asarray(take, np.int64)

    def __iter__(self):
        """One batch ahead: batch k + 1 is sampled -- on a stream of its own when the device is a GPU -- while the consumer
        works on batch k; the consumer's stream waits for the batch's event, nothing waits for the consumer."""
        rows = self._row_batches()
        if self.device.type != "cuda":
            for r in rows:
                yield self._device_batch(r) if self.one_pos else self._device_batch_prop(r)
            return
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))   # the uploads of __init__ (rows, CSR) come first

        def produce():
            with torch.cuda.stream(side):
                r = next(rows)
                b = self._device_batch(r) if self.one_pos else self._device_batch_prop(r)
                ev = torch.cuda.Event()
                ev.record(side)
            return b, ev

        nxt = produce()
        while True:
            b, ev = nxt
            nxt = produce()
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for v in b.values():
                v.record_stream(cur)          # allocated on the side stream, consumed on the caller's
            yield b


class OneVsAllTrainDataset(object):
    """1-vs-all training batches: `train_dataset(..., num_labels=None)` (data.py:157-158 -> `_add_lookup_values`, :314-330;
    selected by an empty `num_labels` in `config_nations_plain.yaml:22`, `config_umls_cpg.yaml:20`; `run_cpg.py:116` then builds the
    model with `use_negative_sampling=False`).  One row per (e1, rel) RECORD of the train graph:
        e1, rel int64 [B];  e2 = -1 (the train records carry 'None', data.py:485-489 -> `entity_ids['None'] = -1`, :333);
        e2_multi float32 [B, |E|]: 1.0 at every known train tail of (e1, rel) (`tf.sparse_to_dense`, data.py:318-322);
        lookup_values int32 [B, 0] (data.py:323).
    Rows pass through the shuffle buffer of 1000 (data.py:160) and are batched; the record stream repeats (data.py:136).
    `device`: the dense label matrix is built THERE from the CSR tail lists (one scatter of the batch's known tails into a zeroed
    [B, |E|] tensor -- 30 MB per 512 x 14,541 batch never crosses PCIe) and the batch is a dict of device tensors that
    `ConvE.train_step` takes as it is; without it, NumPy arrays in the reference's dtypes."""

    def __init__(self, samples, num_ent, batch_size, seed=0, shuffle_buffer=1000, device=None):
        self.num_ent, self.batch_size, self.shuffle_buffer = int(num_ent), int(batch_size), int(shuffle_buffer)
        self.rng = np.random.default_rng(seed)
        self.e1 = np.asarray(samples["e1"], np.int64)
        self.rel = np.asarray(samples["rel"], np.int64)
        self.ip = np.asarray(samples["tail_indptr"], np.int64)
        self.tails = np.asarray(samples["tail_idx"], np.int64)
        self.n_rec = len(self.e1)
        if self.n_rec == 0:
            raise ValueError("OneVsAllTrainDataset: the train graph has no (e1, rel) record")
        self.device = torch.device(device) if device is not None else None
        if self.device is not None:
            self.d_e1 = torch.as_tensor(self.e1).to(self.device)
            self.d_rel = torch.as_tensor(self.rel).to(self.device)
            self.d_ip = torch.as_tensor(self.ip).to(self.device)
            self.d_tails = torch.as_tensor(self.tails).to(self.device)

    def _record_batches(self):
        buf, pos = [], 0
        while True:
            while len(buf) < self.shuffle_buffer + self.batch_size:
                buf.append(pos % self.n_rec)
                pos += 1
            js = self.rng.integers(0, 1 << 62, size=self.batch_size)
            take = []
            for t in range(self.batch_size):
                j = int(js[t] % min(len(buf), self.shuffle_buffer))
                take.append(buf[j])
                buf[j] = buf[-1]
                buf.pop()
            yield np.asarray(take, np.int64)

    def _host_batch(self, rec):
        B = len(rec)
        lab = np.zeros((B, self.num_ent), np.float32)
        for i, r in enumerate(rec):
            lab[i, self.tails[self.ip[r]:self.ip[r + 1]]] = 1.0
        return dict(e1=self.e1[rec], rel=self.rel[rec], e2=np.full(B, -1, np.int64), e2_multi=lab,
                    lookup_values=np.zeros((B, 0), np.int32))

    def _device_batch(self, rec):
        dev = self.device
        B = len(rec)
        r = torch.as_tensor(rec).to(dev)
        lo, hi = self.d_ip[r], self.d_ip[r + 1]
        n = hi - lo
        owner = torch.repeat_interleave(torch.arange(B, device=dev), n)
        base = torch.cumsum(n, 0) - n
        src = lo[owner] + (torch.arange(int(owner.numel()), device=dev) - base[owner])
        lab = torch.zeros((B, self.num_ent), dtype=torch.float32, device=dev)
        lab[owner, self.d_tails[src]] = 1.0
        return dict(e1=self.d_e1[r], rel=self.d_rel[r], e2=torch.full((B,), -1, dtype=torch.int64, device=dev), e2_multi=lab,
                    lookup_values=torch.zeros((B, 0), dtype=torch.int32, device=dev))

    def __iter__(self) -> Iterator[dict]:
        for rec in self._record_batches():
            yield self._host_batch(rec) if self.device is None else self._device_batch(rec)


class TrainDataset(object):
    """Endless training-batch source in the reference batch contract (models.py:139-152): the mirror of
    `train_dataset` (data.py:89-166) with its two samplers restated in NumPy.

    `samples`: dict with `e1, rel` int64 [N] and the CSR list of ALL known tails of each (e1, rel) in the train
    graph (`tail_indptr [N+1]`, `tail_idx`) -- what one TFRecord of the reference holds (data.py:574-594).

    * one_positive_label_per_sample=True (the default, data.py:138-144 + :278-311): every (e1, rel, e2_i) becomes
      its own row; lookup = [e2_i, a window of num_labels-1 consecutive entries of a fresh permutation of the
      entities starting at a uniform offset]; labels = membership of the looked-up ids in the tail list (a sampled
      "negative" that is a known tail is supervised as positive, as the reference comments).
    * otherwise (data.py:228-277): positives shuffled, negatives a prefix of a permutation;
      num_positives_needed = int(num_labels / (1 + prop_negatives)).
    Rows pass through a shuffle buffer of 1000 (data.py:160) and are batched.  TF's RNG streams cannot be
    reproduced: parity with the reference is distributional (tests check the construction rules)."""

    def __init__(self, samples, num_ent, batch_size, num_labels=100, one_positive_label_per_sample=True,
                 prop_negatives=10.0, seed=0, shuffle_buffer=1000):
        self.s, self.num_ent, self.batch_size = samples, int(num_ent), int(batch_size)
        self.num_labels, self.one_pos, self.prop = int(num_labels), bool(one_positive_label_per_sample), float(prop_negatives)
        self.shuffle_buffer = int(shuffle_buffer)
        self.rng = np.random.default_rng(seed)
        if self.num_labels > self.num_ent:
            raise ValueError("num_labels needs to be at most the total number of entities (data.py:146-147)")

    # one TFRecord -> rows (lookup [k, L], labels [k, L], e1, rel, e2)
    def _rows_one_positive(self, i):
        s, L, E = self.s, self.num_labels, self.num_ent
        tails = s["tail_idx"][s["tail_indptr"][i]:s["tail_indptr"][i + 1]]
        k = len(tails)
        perm = self.rng.permutation(E)
        start = self.rng.integers(0, E - (L - 1), size=k) if E - (L - 1) > 0 else np.zeros(k, np.int64)
        neg = perm[start[:, None] + np.arange(L - 1)[None, :]]
        lookup = np.concatenate([tails[:, None], neg], axis=1)
        member = np.zeros(E, np.float32)
        member[tails] = 1.0
        return lookup, member[lookup], np.full(k, s["e1"][i]), np.full(k, s["rel"][i]), tails

    def _row_prop_negatives(self, i):
        s, L, E = self.s, self.num_labels, self.num_ent
        tails = self.rng.permutation(s["tail_idx"][s["tail_indptr"][i]:s["tail_indptr"][i + 1]])
        wrong = self.rng.permutation(E)
        need = int(1.0 / (1.0 + self.prop) * L)
        if len(tails) <= need:
            idx = np.concatenate([tails, wrong[:L - len(tails)]])
        else:
            n_neg = min(E, L - need)
            idx = np.concatenate([tails[:L - n_neg], wrong[:n_neg]])
        member = np.zeros(E, np.float32)
        member[tails] = 1.0
        e2 = tails[0] if len(tails) else -1
        return idx[None, :], member[idx][None, :], np.array([s["e1"][i]]), np.array([s["rel"][i]]), np.array([e2])

    def __iter__(self) -> Iterator[dict]:
        N = len(self.s["e1"])
        buf = []          # shuffle buffer of rows
        order = np.arange(N)
        pos = 0
        while True:
            while len(buf) < self.shuffle_buffer + self.batch_size:
                i = order[pos % N]
                pos += 1                                   # .repeat(): the record stream wraps around
                lk, lab, e1, rel, e2 = self._rows_one_positive(i) if self.one_pos else self._row_prop_negatives(i)
                for j in range(len(e1)):
                    buf.append((lk[j], lab[j], e1[j], rel[j], e2[j]))
            take = []
            for _ in range(self.batch_size):
                j = int(self.rng.integers(0, min(len(buf), self.shuffle_buffer)))
                take.append(buf[j])
                buf[j] = buf[-1]
                buf.pop()
            yield dict(e1=np.array([t[2] for t in take], np.int64), rel=np.array([t[3] for t in take], np.int64),
                       e2=np.array([t[4] for t in take], np.int64),
                       lookup_values=np.stack([t[0] for t in take]).astype(np.int32),
                       e2_multi=np.stack([t[1] for t in take]).astype(np.float32))
