"""TSV knowledge-graph loader: the on-disk side of the reference's data pipeline,
`qa_cpg/data.py:401-572`, producing the batch contract of `coper_amd.data` (ids + CSR filters)
instead of TFRecords and dense masks.

What is kept exactly as the reference does it:
  * triples `e1\\trel\\te2` per line, fields stripped (data.py:417-421);
  * every triple also feeds the `_reverse` relation `(e2, rel_reverse) -> e1` of the FULL graph
    (data.py:422-437), and of a split's own graph only where `add_reverse_per_filetype` says so
    (data.py:438-439: train only for the shipped loaders);
  * evaluation labels / filters come from the full graph (train + dev + test, both directions)
    (data.py:464-469, 494-503): the filter of a query (e1, rel) is every known tail;
  * `needs_test_set_cleaning`: dev/test questions whose e1, e2 or relation never occur in train are
    dropped (data.py:445-459, 487-497);
  * JSON lines `{"e1","e2","rel","e2_multi"}` with space-joined tails and e2 = "None" for the train
    file (data.py:477-504), `entities.txt` / `relations.txt` id maps that, once written, define the
    ids (data.py:506-572);
  * eval batches exclude inverse relations unless asked (data.py:192-197, run_cpg.py:156).

What differs on purpose: id assignment when no id files exist iterates sorted names instead of Python
`set`s (the reference's ids depend on the hash seed), and nothing is downloaded."""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Tuple

import numpy as np

from .data import DeviceTrainDataset, EvalDataset, OneVsAllTrainDataset, TrainDataset

__all__ = ["TSVKGLoader"]


class TSVKGLoader(object):
    def __init__(self, directory: str, dataset_name: str = "kg", filetypes=("train", "dev", "test"),
                 add_reverse_per_filetype=(True, False, False), needs_test_set_cleaning: bool = False):
        self.directory, self.dataset_name = directory, dataset_name
        self.filetypes = list(filetypes)
        self.add_reverse_per_filetype = list(add_reverse_per_filetype)
        self.needs_test_set_cleaning = needs_test_set_cleaning
        self.num_ent: Optional[int] = None
        self.num_rel: Optional[int] = None
        self.full_graph: Dict[Tuple[str, str], set] = {}
        self.graphs: Dict[str, Dict[Tuple[str, str], set]] = {}
        self.entity_ids: Dict[str, int] = {}
        self.relation_ids: Dict[str, int] = {}
        self._samples: Dict[str, List[dict]] = {}
        self._loaded = False

    # ---------------------------------------------------------------- data.py:401-475
    def load_and_preprocess(self, write_json: bool = False):
        full_graph, graphs = {}, {}
        for i, ft in enumerate(self.filetypes):
            g = graphs[ft] = {}
            with open(os.path.join(self.directory, "%s.txt" % ft), "r") as handle:
                for line in handle:
                    if not line.strip():
                        continue
                    e1, rel, e2 = line.split("\t")
                    e1, rel, e2 = e1.strip(), rel.strip(), e2.strip()
                    rel_reverse = rel + "_reverse"
                    full_graph.setdefault((e1, rel), set()).add(e2)
                    full_graph.setdefault((e2, rel_reverse), set()).add(e1)
                    g.setdefault((e1, rel), set()).add(e2)
                    g.setdefault((e2, rel_reverse), set())
                    if self.add_reverse_per_filetype[i]:
                        g[(e2, rel_reverse)].add(e1)
        self.full_graph, self.graphs = full_graph, graphs
        allowed_entities = allowed_relations = None
        if self.needs_test_set_cleaning:
            allowed_entities, allowed_relations = set(), set()
            for (e1, rel), tails in graphs[self.filetypes[0]].items():
                allowed_entities.add(e1)
                allowed_entities.update(tails)
                allowed_relations.add(rel)
        names = {"train": self.filetypes[0], "dev": self.filetypes[1], "test": self.filetypes[2]}
        self._samples = {
            "train": self._graph_samples(graphs[names["train"]], None, None, None),
            "dev": self._graph_samples(graphs[names["dev"]], full_graph, allowed_entities, allowed_relations),
            "test": self._graph_samples(graphs[names["test"]], full_graph, allowed_entities, allowed_relations),
            "full": self._graph_samples(full_graph, full_graph, allowed_entities, allowed_relations),
        }
        if write_json:
            for split, samples in self._samples.items():
                with open(os.path.join(self.directory, "e1rel_to_e2_%s.json" % split), "w") as handle:
                    for s in samples:
                        handle.write(json.dumps({"e1": s["e1"], "e2": s["e2"], "rel": s["rel"],
                                                 "e2_multi": " ".join(s["e2_multi"])}) + "\n")
        self._loaded = True
        return self._samples

    @staticmethod
    def _graph_samples(graph, labels, allowed_entities, allowed_relations):
        """data.py:477-504 (`_write_graph`) as in-memory samples; tails kept sorted for determinism."""
        out = []
        for (e1, rel), value in graph.items():
            if labels is None:
                out.append({"e1": e1, "e2": "None", "rel": rel, "e2_multi": sorted(value)})
                continue
            if allowed_entities is not None and e1 not in allowed_entities:
                continue
            if allowed_relations is not None and rel not in allowed_relations:
                continue
            e2_multi = sorted(labels[(e1, rel)])
            for e2 in sorted(value):
                if allowed_entities is not None and e2 not in allowed_entities:
                    continue
                out.append({"e1": e1, "e2": e2, "rel": rel, "e2_multi": e2_multi})
        return out

    # ---------------------------------------------------------------- data.py:506-572
    def assign_ids(self, write_files: bool = False):
        if not self._loaded:
            self.load_and_preprocess()
        ent_file = os.path.join(self.directory, "entities.txt")
        rel_file = os.path.join(self.directory, "relations.txt")
        entity_ids, relation_ids = {}, {}
        if os.path.exists(ent_file):
            with open(ent_file) as handle:
                for i, line in enumerate(handle):
                    entity_ids[line.strip()] = i
        if os.path.exists(rel_file):
            with open(rel_file) as handle:
                for i, line in enumerate(handle):
                    relation_ids[line.strip()] = i
        if not entity_ids or not relation_ids:
            ents, rels = set(), set()
            for s in self._samples["full"]:
                ents.add(s["e1"]); ents.add(s["e2"]); ents.update(s["e2_multi"])
                rels.add(s["rel"])
            ents.discard("None"); rels.discard("None")
            if not entity_ids:
                entity_ids = {name: i for i, name in enumerate(sorted(ents))}
                if write_files:
                    with open(ent_file, "w") as handle:
                        handle.write("".join(n + "\n" for n in sorted(ents)))
            if not relation_ids:
                relation_ids = {name: i for i, name in enumerate(sorted(rels))}
                if write_files:
                    with open(rel_file, "w") as handle:
                        handle.write("".join(n + "\n" for n in sorted(rels)))
        self.entity_ids, self.relation_ids = entity_ids, relation_ids
        self.num_ent, self.num_rel = len(entity_ids), len(relation_ids)    # data.py:337-338
        return entity_ids, relation_ids

    def maybe_create_tf_record_files(self, directory=None, buffer_size=None, write_tfrecords=False,
                                     max_records_per_file=1000000):
        """Name kept for the driver (run_cpg.py:108): loads and assigns ids.  This engine feeds from the in-memory
        id arrays; `write_tfrecords=True` also writes the reference's `<split>-<n>.tfrecords` files
        (data.py:353-388) -- only the missing splits, as the reference does -- so that its own loader finds them.
        Returns {split: [file names]} of what exists / was written, or None."""
        self.assign_ids()
        if not write_tfrecords:
            return None
        import glob
        from . import tf_records
        directory = directory or self.directory
        E, Rm = self.entity_ids, self.relation_ids
        out = {}
        for split in ("train", "dev", "test"):
            existing = glob.glob(os.path.join(directory, "%s-*.tfrecords" % split))
            if existing:                       # data.py:360-363: not recreated
                out[split] = sorted(existing)
                continue
            samples = ({"e1": E.get(s["e1"], -1), "e2": E.get(s["e2"], -1), "rel": Rm.get(s["rel"], -1),
                        "e2_multi": [E[t] for t in s["e2_multi"] if t != "None"],
                        "is_inverse": s["rel"].endswith("_reverse")} for s in self._samples[split])
            out[split] = tf_records.write_split(directory, split, samples, max_records_per_file)
        return out

    # ---------------------------------------------------------------- data.py:168-226 + 574-594
    def encoded_split(self, dataset_type: str, include_inv_relations: bool = False):
        """Id arrays + CSR filter of one split: what `_encode_sample_as_tf_record` + `eval_dataset` feed."""
        if not self.entity_ids:
            self.assign_ids()
        E, Rm = self.entity_ids, self.relation_ids
        e1, e2, rel, indptr, idx = [], [], [], [0], []
        for s in self._samples[dataset_type]:
            if s["e2"] == "None":
                continue
            if s["rel"].endswith("_reverse") and not include_inv_relations:
                continue
            e1.append(E[s["e1"]]); e2.append(E[s["e2"]]); rel.append(Rm[s["rel"]])
            tails = sorted(set(E[t] for t in s["e2_multi"] if t != "None"))
            idx.extend(tails)
            indptr.append(len(idx))
        return dict(e1=np.asarray(e1, np.int64), e2=np.asarray(e2, np.int64), rel=np.asarray(rel, np.int64),
                    filt_indptr=np.asarray(indptr, np.int64), filt_idx=np.asarray(idx, np.int64))

    def eval_dataset(self, directory=None, dataset_type="test", batch_size=512, include_inv_relations=False,
                     buffer_size=None, prefetch_buffer_size=None, dense_mask=False):
        if self.num_ent is None:
            self.assign_ids()
        return EvalDataset(self.encoded_split(dataset_type, include_inv_relations), batch_size, self.num_ent, dense_mask)

    # ---------------------------------------------------------------- data.py:89-166
    def train_samples(self, include_inv_relations: bool = True):
        """One record per (e1, rel) of the train graph with the list of all its train tails -- the content of the
        reference's train TFRecords (`_graph_samples(..., labels=None)`, data.py:481-484)."""
        if not self.entity_ids:
            self.assign_ids()
        E, Rm = self.entity_ids, self.relation_ids
        e1, rel, indptr, idx = [], [], [0], []
        for s in self._samples["train"]:
            if not s["e2_multi"]:
                continue
            if s["rel"].endswith("_reverse") and not include_inv_relations:
                continue
            e1.append(E[s["e1"]]); rel.append(Rm[s["rel"]])
            idx.extend(sorted(set(E[t] for t in s["e2_multi"] if t != "None")))
            indptr.append(len(idx))
        return dict(e1=np.asarray(e1, np.int64), rel=np.asarray(rel, np.int64), tail_indptr=np.asarray(indptr, np.int64),
                    tail_idx=np.asarray(idx, np.int64))

    def train_dataset(self, directory=None, batch_size=512, include_inv_relations=True, num_parallel_readers=None,
                      num_parallel_batches=None, buffer_size=None, prefetch_buffer_size=None, prop_negatives=10.0,
                      num_labels=100, cache=False, one_positive_label_per_sample=True, seed=0, device=None):
        """`device` (extra keyword): sample on that device (`DeviceTrainDataset`: batches of device tensors; the
        both samplers) instead of on the host."""
        if num_labels is None:      # 1-vs-all labels (data.py:157-158, 314-330)
            return OneVsAllTrainDataset(self.train_samples(include_inv_relations), self.num_ent, batch_size, seed, device=device)
        if device is not None:
            return DeviceTrainDataset(self.train_samples(include_inv_relations), self.num_ent, batch_size, num_labels, seed, device=device,
                                      one_positive_label_per_sample=one_positive_label_per_sample, prop_negatives=prop_negatives)
        return TrainDataset(self.train_samples(include_inv_relations), self.num_ent, batch_size, num_labels,
                            one_positive_label_per_sample, prop_negatives, seed)


class TFRecordKGLoader(object):
    """The same loader surface fed from a directory the REFERENCE already preprocessed (`<split>-<n>.tfrecords`,
    `entities.txt`, `relations.txt`; data.py:353-397,506-572): no TSV needed, no TensorFlow needed.  Ids are the
    ones stored in the records, so weights trained by the reference on that directory line up."""

    def __init__(self, directory, dataset_name="tfrecords", num_ent=None, num_rel=None, needs_test_set_cleaning=False):
        self.directory, self.dataset_name = directory, dataset_name
        self.needs_test_set_cleaning = needs_test_set_cleaning
        self.num_ent, self.num_rel = num_ent, num_rel

    def maybe_create_tf_record_files(self, directory=None, buffer_size=None):
        """run_cpg.py:108: after this call `num_ent` / `num_rel` are valid (data.py:337-338: the sizes of the id maps)."""
        directory = directory or self.directory
        for attr, fname in (("num_ent", "entities.txt"), ("num_rel", "relations.txt")):
            path = os.path.join(directory, fname)
            if getattr(self, attr) is None:
                if not os.path.exists(path):
                    raise FileNotFoundError("%s not found and %s not given" % (path, attr))
                with open(path) as handle:
                    setattr(self, attr, sum(1 for line in handle if line.strip()))
        return None

    def eval_dataset(self, directory=None, dataset_type="test", batch_size=512, include_inv_relations=False,
                     buffer_size=None, prefetch_buffer_size=None, dense_mask=False):
        from . import tf_records
        if self.num_ent is None:
            self.maybe_create_tf_record_files(directory)
        q = tf_records.read_split(directory or self.directory, dataset_type, include_inv_relations)
        keep = q["e2"] >= 0                                   # samples without a target (e2 'None' -> -1) are not queries
        if not keep.all():
            sel = np.nonzero(keep)[0]
            rows = [q["filt_idx"][q["filt_indptr"][i]:q["filt_indptr"][i + 1]] for i in sel]
            q = dict(e1=q["e1"][sel], e2=q["e2"][sel], rel=q["rel"][sel],
                     filt_indptr=np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64),
                     filt_idx=np.concatenate(rows).astype(np.int64) if rows else np.zeros(0, np.int64))
        return EvalDataset(q, batch_size, self.num_ent, dense_mask)

    def train_dataset(self, directory=None, batch_size=512, include_inv_relations=True, num_parallel_readers=None,
                      num_parallel_batches=None, buffer_size=None, prefetch_buffer_size=None, prop_negatives=10.0,
                      num_labels=100, cache=False, one_positive_label_per_sample=True, seed=0, device=None):
        from . import tf_records
        if self.num_ent is None:
            self.maybe_create_tf_record_files(directory)
        q = tf_records.read_split(directory or self.directory, "train", include_inv_relations)
        n = np.diff(q["filt_indptr"])
        sel = np.nonzero(n > 0)[0]
        rows = [q["filt_idx"][q["filt_indptr"][i]:q["filt_indptr"][i + 1]] for i in sel]
        samples = dict(e1=q["e1"][sel], rel=q["rel"][sel],
                       tail_indptr=np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64),
                       tail_idx=np.concatenate(rows).astype(np.int64) if rows else np.zeros(0, np.int64))
        if num_labels is None:      # 1-vs-all labels (data.py:157-158, 314-330)
            return OneVsAllTrainDataset(samples, self.num_ent, batch_size, seed, device=device)
        if device is not None:
            return DeviceTrainDataset(samples, self.num_ent, batch_size, num_labels, seed, device=device,
                                      one_positive_label_per_sample=one_positive_label_per_sample, prop_negatives=prop_negatives)
        return TrainDataset(samples, self.num_ent, batch_size, num_labels, one_positive_label_per_sample, prop_negatives, seed)
