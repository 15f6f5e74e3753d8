"""`ranking_and_hits` -- drop-in for `qa_cpg.metrics.ranking_and_hits` (metrics.py:23-86).

Same signature, same return `(mr, mrr, hits)`, same float64 means (metrics.py:65-76), same
optional result files (metrics.py:14-20,70-83).  What changes is where the work happens: the
reference fetches `[B,|E|]` logits and a dense `[B,|E|]` filter mask to the host every batch and
argsorts each row in Python (metrics.py:40-57); here the batches are concatenated, encoded and
ranked on the device (fused score + count, CSR filter), and only int32 ranks come back."""
from __future__ import annotations

import logging
import os

import numpy as np
import torch

from .data import canonical_csr, dense_filter_to_csr
from .sharding import local_rank_pass

__all__ = ["ranking_and_hits", "hits_and_means", "collect_batches"]

logger = logging.getLogger(__name__)


def _write_data_to_file(file_path, data):
    append_write = "a" if os.path.exists(file_path) else "w+"
    with open(file_path, append_write) as handle:
        handle.write(str(data) + "\n")


def hits_and_means(ranks, hits_to_compute=(1, 3, 5, 10, 20)):
    """metrics.py:53-57,65-76: per-query 1.0/0.0 hits, float64 means.  (The mean of 1.0 / 0.0 flags is their count over their
    number, exactly: a sum of ones is exact in float64 whatever the order -- np.float64(count) / n is the value
    np.mean(np.where(ranks <= k, 1.0, 0.0)) has, at a third of the passes over the array.)"""
    ranks = np.asarray(ranks)
    if ranks.dtype.kind not in "iu":
        ranks = ranks.astype(np.int64)
    n = len(ranks)
    hits = {}
    levels = [int(k) for k in hits_to_compute]
    # int32 ranks as a pass delivers them: ONE pass over the array in the library's host code (coper_hits_means: the integer sum,
    # the float64 quotients summed in np.mean's own pairwise order, a counter per level) instead of a dozen NumPy calls of
    # 3 - 5 us each -- 50 -> 12 us per 20,480 ranks; anything it refuses (a rank below 1) takes the NumPy route below
    if n and ranks.dtype == np.int32 and ranks.ndim == 1 and ranks.flags.c_contiguous and len(levels) <= 16 \
            and all(-2 ** 31 <= k < 2 ** 31 for k in levels):
        out = _hits_means_native(ranks, levels)
        if out is not None:
            mr, mrr, hv = out
            return mr, mrr, {k: hv[i] for i, k in enumerate(hits_to_compute)}
    if n and levels and 0 < max(levels) <= 4096 and int(ranks.min()) >= 0:
        # every level from ONE pass over the array: the ranks up to the largest level are picked out (a few percent of them),
        # counted per value, and the counts accumulated
        top = max(levels)
        cum = np.cumsum(np.bincount(ranks[ranks <= top], minlength=top + 1))
        for hits_level in hits_to_compute:
            k = int(hits_level)
            hits[hits_level] = np.float64(cum[k] if k >= 0 else 0) / n
    else:
        for hits_level in hits_to_compute:
            hits[hits_level] = np.float64(np.count_nonzero(ranks <= hits_level)) / n if n else float("nan")
    # (np.mean of an integer array is the float64 sum over n; the sum of integers below 2^53 is exact in any order, so the
    # integer sum gives the same float64 -- at half the time)
    mr = (np.float64(ranks.sum(dtype=np.int64)) / n if ranks.dtype.kind in "iu" and ranks.dtype.itemsize <= 4 else np.mean(ranks)) if n else float("nan")
    if n and ranks.dtype.kind in "iu" and int(ranks.min()) >= 1 and int(ranks.max()) < (1 << 22):
        # 1 / rank from a table (the same float64 quotients, gathered in the same order: np.mean sums the same array) -- a third of
        # the time of 20,480 int -> float64 conversions and divisions
        mrr = np.mean(_inverse_table(int(ranks.max()))[ranks])
    else:
        mrr = np.mean(1. / ranks) if n else float("nan")
    return mr, mrr, hits


def _hits_means_native(ranks, levels):
    import ctypes as C
    try:
        from . import _lib
        lib = _lib.load()
    except Exception:       # (the metrics of host ranks do not need the library: NumPy route)
        return None
    lv = (C.c_int32 * max(1, len(levels)))(*levels)
    hv = (C.c_double * max(1, len(levels)))()
    mr, mrr = C.c_double(), C.c_double()
    if lib.coper_hits_means(C.c_void_p(ranks.ctypes.data), len(ranks), lv, len(levels), C.byref(mr), C.byref(mrr), hv) != 0:
        return None
    return np.float64(mr.value), np.float64(mrr.value), [np.float64(hv[i]) for i in range(len(levels))]


_INV = np.zeros(1)


def _inverse_table(top):
    """[0, 1/1, 1/2, ... 1/top] in float64 (grown on demand, kept)."""
    global _INV
    if len(_INV) <= top:
        n = max(top + 1, 2 * len(_INV))
        t = np.zeros(n)
        t[1:] = 1. / np.arange(1, n)
        _INV = t
    return _INV


def collect_batches(data_iterator_handle, device=None):
    """Drains a batch source (the role of the `while not stopped` loop, metrics.py:38-60) into one set
    of query arrays with a CSR filter.  Accepts batches carrying `filt_indptr`/`filt_idx` or the
    reference's dense `e2_multi`."""
    if hasattr(data_iterator_handle, "as_single_batch"):
        return data_iterator_handle.as_single_batch()
    e1, e2, rel, indptr, idx = [], [], [], [np.zeros(1, np.int64)], []
    base = 0
    for batch in data_iterator_handle:
        n = len(batch["e1"])
        if n == 0:
            continue
        e1.append(np.asarray(batch["e1"], np.int64))
        e2.append(np.asarray(batch["e2"], np.int64))
        rel.append(np.asarray(batch["rel"], np.int64))
        if "filt_indptr" in batch:
            ip, ix = np.asarray(batch["filt_indptr"], np.int64), np.asarray(batch["filt_idx"], np.int64)
        else:
            ip, ix = dense_filter_to_csr(np.asarray(batch["e2_multi"]), device=device)   # on the model's device when it has one
        indptr.append(ip[1:] + base)
        idx.append(ix)
        base += int(ip[-1])
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    # the rank kernels take every filter row sorted ascending: one vectorised check over the whole pass (rows that come
    # unsorted are sorted)
    ip_all, ix_all = canonical_csr(np.concatenate(indptr), cat(idx, np.int64))
    return dict(e1=cat(e1, np.int64), e2=cat(e2, np.int64), rel=cat(rel, np.int64), filt_indptr=ip_all, filt_idx=ix_all)


def ranking_and_hits(model, results_dir, data_iterator_handle, name, session=None, hits_to_compute=(1, 3, 5, 10, 20),
                     enable_write_to_file=False, max_chunk=32768, ranker=None, return_ranks=False):
    """`session` is accepted and ignored (the model object is the session).  Extra keyword-only
    options: `max_chunk` queries per device pass, `ranker` = an `EntityShardedRanker` for
    entity-sharded multi-GPU evaluation, `return_ranks` to also get the int ranks."""
    if results_dir is not None:
        os.makedirs(results_dir, exist_ok=True)
    logger.info("")
    logger.info("-" * 50)
    logger.info(name)
    logger.info("-" * 50)
    logger.info("")

    # an evaluation set that is scored again and again (after every epoch, run_cpg.py:228-250) is marshalled once: int32 in
    # pinned memory, canonical CSR, device buffers (EvalDataset.staged_for); a pass is then three asynchronous calls and one wait
    # Sets larger than one device pass (`max_chunk` queries; FB15k-237's test set with both directions is 40,932) are cut into equal
    # chunks, each staged once; their passes are queued back to back on one stream -- the next chunk's sort by relation inside the
    # running chunk's encoder launch (coper_group_next: its ids are resident) -- and waited for once.
    sbs = None
    if ranker is None and hasattr(model, "rank_pass_staged"):
        if hasattr(data_iterator_handle, "staged_chunks_for"):
            sbs = data_iterator_handle.staged_chunks_for(model, max_chunk)
        elif hasattr(data_iterator_handle, "staged_for") and getattr(data_iterator_handle, "num_queries", max_chunk + 1) <= max_chunk:
            sb = data_iterator_handle.staged_for(model)
            sbs = [sb] if sb is not None else None
    if sbs:
        Q = sum(sb["B"] for sb in sbs)
        x3 = Q and getattr(model, "score_mode", None) == "bf16x3"
        todo = [i for i, sb in enumerate(sbs) if sb["B"]]
        for attempt in range(_MAX_RERANKS + 1):
            for j, i in enumerate(todo):
                nxt = sbs[todo[j + 1]] if j + 1 < len(todo) else None
                if nxt is not None and nxt.get("resident") and hasattr(model, "group_next"):
                    model.group_next(nxt["views"][0], nxt["views"][1])
                model.rank_pass_staged(sbs[i])
            if todo:
                sbs[todo[-1]]["event"].synchronize()          # (one stream: the last chunk's event covers them all)
            again = []
            if x3:
                # ONE policy call per round of passes: every chunk of the round was audited under the same kappa, and
                # coper_band_policy multiplies the handle's kappa cumulatively -- per chunk, two chunks at ratio 1.2 widened 8 x 8.
                # The largest ratio and the summed pairs decide; the chunks whose own ratio reached 1 are ranked again.
                auds = []
                for i in todo:
                    aud = sbs[i]["out_host"][sbs[i]["B"]:sbs[i]["B"] + 2].numpy()
                    auds.append((i, float(aud[:1].view(np.float32)[0]), int(aud[1:2].view(np.uint32)[0])))
                seen = [a for a in auds if a[2] > 0]
                if seen and _act_on_band_audit(model, max(a[1] for a in seen), sum(a[2] for a in seen)) == 2:
                    again = [a[0] for a in seen if a[1] >= 1.0]
            # the guard of a grouping prepared ahead (include/coper_hip.h: coper_group_next): a pass whose ids were rewritten between
            # the launch that sorted them and the pass itself comes back with COPER_RANK_STALE in every rank -- ranked again
            for i in todo:
                if i not in again and int(sbs[i]["out_host"][:sbs[i]["B"]].min()) < 1:
                    again.append(i)
                    ranking_and_hits.stale_passes = getattr(ranking_and_hits, "stale_passes", 0) + 1
            again.sort()
            if not again:
                break
            todo = again
        else:
            raise RuntimeError("bf16x3 band audit: still above the band's allowance after %d re-ranked passes" % _MAX_RERANKS)
        parts = [sb["out_host"][:sb["B"]].numpy() for sb in sbs]
        ranks = parts[0] if len(parts) == 1 else np.concatenate(parts)
        return _finish(ranks, Q, results_dir, hits_to_compute, enable_write_to_file, return_ranks)

    # a plain list of batches with CSR filters: the encoder needs the ids only, so it is launched as soon as THEY are
    # concatenated, and the filters -- the bulk of the marshalling: concatenation, the canonical-order check, staging -- are
    # put together on the host while the device encodes (coper_encode + coper_rank give coper_encode_rank's bits)
    # An iterator of batches is drained into a list only when its batches carry CSR filters (ids and short lists: a few MB for a
    # whole evaluation set).  The reference's own batches carry a dense e2_multi [B, |E|] (30 MB each, 1.2 GB for the FB15k-237
    # test set): those are streamed through collect_batches below, one batch converted to CSR and dropped at a time.
    if not hasattr(data_iterator_handle, "as_single_batch") and not isinstance(data_iterator_handle, (list, tuple)):
        it = iter(data_iterator_handle)
        first = next(it, None)
        if first is None:
            data_iterator_handle = []
        elif "filt_indptr" in first:
            data_iterator_handle = [first] + list(it)
        else:
            import itertools
            data_iterator_handle = itertools.chain([first], it)
    if ranker is None and isinstance(data_iterator_handle, (list, tuple)) and hasattr(model, "encode") and hasattr(model, "rank") \
            and 0 < sum(len(b["e1"]) for b in data_iterator_handle) <= max_chunk \
            and all("filt_indptr" in b for b in data_iterator_handle):
        bs = [b for b in data_iterator_handle if len(b["e1"])]
        e1 = np.concatenate([np.asarray(b["e1"], np.int64) for b in bs])
        rel = np.concatenate([np.asarray(b["rel"], np.int64) for b in bs])
        h = model.encode(e1, rel)                                          # asynchronous
        e2 = np.concatenate([np.asarray(b["e2"], np.int64) for b in bs])
        nnzs = [int(np.asarray(b["filt_indptr"])[-1]) for b in bs]
        # the row pointers: the batches' own (without their leading zeros) in one concatenation, each batch's offset added in one
        # vectorised pass (40 small additions before)
        ip = np.empty(len(e2) + 1, np.int64)
        ip[0] = 0
        np.concatenate([np.asarray(b["filt_indptr"], np.int64)[1:] for b in bs], out=ip[1:])
        if len(bs) > 1:
            ip[1:] += np.repeat(np.cumsum([0] + nnzs[:-1]), [len(b["e1"]) for b in bs])
        ix = np.concatenate([np.asarray(b["filt_idx"], np.int64) for b in bs]) if ip[-1] else np.zeros(0, np.int64)
        # targets + filter into the staging buffer with the checks in the same native pass (ConvE.stage_csr); rows that come
        # unsorted, or ids beyond int32, take the general route
        staged = model.stage_csr(e2, ip, ix) if hasattr(model, "stage_csr") else None
        if staged is not None:
            e2, ip, ix_arg = staged
            nnz = len(ix)
        else:
            ip, ix = canonical_csr(ip, ix)
            ix_arg, nnz = ix, len(ix)
        x3 = getattr(model, "score_mode", None) == "bf16x3" and hasattr(model, "band_audit")
        for attempt in range(_MAX_RERANKS + 1):
            r, _ = model.rank(h, e2, ip, ix_arg, filt_nnz=nnz, want_equal=False)
            if hasattr(model, "fetch_ranks_audit"):        # ranks + the audit's words: one launch, one wait (pinned)
                ranks, ratio, pairs = model.fetch_ranks_audit(r)
                if not x3 or _act_on_band_audit(model, ratio, pairs) != 2:
                    break
                continue
            ranks = r.cpu().numpy()
            if not x3 or _act_on_band_audit(model, *model.band_audit()) != 2:
                break
        else:
            raise RuntimeError("bf16x3 band audit: still above the band's allowance after %d re-ranked passes" % _MAX_RERANKS)
        Q = len(e1)
        return _finish(ranks, Q, results_dir, hits_to_compute, enable_write_to_file, return_ranks)

    q = collect_batches(data_iterator_handle, device=getattr(model, "device", None))
    Q = len(q["e1"])
    ranks = []

    def chunks():
        for s in range(0, Q, max_chunk):
            e = min(Q, s + max_chunk)
            ip = q["filt_indptr"][s:e + 1]
            yield dict(e1=q["e1"][s:e], e2=q["e2"][s:e], rel=q["rel"][s:e], filt_indptr=ip - ip[0], filt_idx=q["filt_idx"][ip[0]:ip[-1]])

    if ranker is not None and hasattr(ranker, "rank_stream"):
        # an entity-sharded ranker owns its loop (sharding.py: plans one chunk ahead, steps 1 - 2 of the next chunk beside this one's
        # count launch, the band audit of every shard handle acted on per window of chunks)
        ranks = [res[0] for res in ranker.rank_stream(chunks())]
    else:
        for chunk in chunks():
            # bf16x3 mode: the run-time audit of the exact band (include/coper_hip.h: coper_band_audit) -- the largest error of the
            # mode's logits on the pairs closest to the targets, relative to what the band allows; ranks are the fp32 chain's below 1.
            # Acted on per chunk (coper_band_policy): above 0.5 the handle's kappa is widened, from 1.0 on the chunk is ranked again.
            x3 = ranker is None and getattr(model, "score_mode", None) == "bf16x3" and hasattr(model, "band_audit")
            for attempt in range(_MAX_RERANKS + 1):
                if ranker is not None:
                    r, _ = ranker.rank(chunk)
                else:
                    r, _ = local_rank_pass(model, chunk)
                if not x3 or _act_on_band_audit(model, *model.band_audit()) != 2:
                    break
            else:
                raise RuntimeError("bf16x3 band audit: still above the band's allowance after %d re-ranked passes" % _MAX_RERANKS)
            ranks.append(r)
    ranks = torch.cat(ranks).cpu().numpy().astype(np.int64) if ranks else np.zeros(0, np.int64)
    return _finish(ranks, Q, results_dir, hits_to_compute, enable_write_to_file, return_ranks)


_MAX_RERANKS = 4


def _act_on_band_audit(model, ratio, n_pairs):
    """The audit ACTS (include/coper_hip.h: coper_band_policy): ratio > 0.5 widens the handle's kappa for later passes, ratio >= 1
    also has the caller rank the pass again (integer ranks are the contract, metrics.py:44-50).  Returns the action.  A pass that
    carried no audit (n_pairs == 0: seven of eight short launches by default) leaves the last real measurement in place."""
    if n_pairs <= 0:
        return 0
    ranking_and_hits.last_band_audit = (ratio, n_pairs)
    action, kappa = model.band_policy(ratio, n_pairs) if hasattr(model, "band_policy") else (0, 0.0)
    if action:
        logger.warning("bf16x3 band audit: |logit_x3 - logit_fp32| reached %.2f of the band's allowance on %d audited pairs; "
                       "kappa widened to %.3g%s", ratio, n_pairs, kappa, "; this pass is ranked again" if action == 2 else "")
        ranking_and_hits.band_actions = getattr(ranking_and_hits, "band_actions", 0) + 1
    return action


def _finish(ranks, count, results_dir, hits_to_compute, enable_write_to_file, return_ranks):
    logger.info("Evaluated %d samples." % count)

    mr, mrr, hits = hits_and_means(ranks, hits_to_compute)
    for hits_level in hits_to_compute:
        logger.info("Hits @%d: %10.6f", hits_level, hits[hits_level])
        if enable_write_to_file:
            _write_data_to_file(os.path.join(results_dir, "hits_at_{}.txt".format(hits_level)), hits[hits_level])
    logger.info("Mean rank: %10.6f", mr)
    logger.info("Mean reciprocal rank: %10.6f", mrr)
    if enable_write_to_file:
        _write_data_to_file(os.path.join(results_dir, "mean_rank.txt"), mr)
        _write_data_to_file(os.path.join(results_dir, "mrr.txt"), mrr)
    logger.info("-" * 50)
    if return_ranks:
        return mr, mrr, hits, np.asarray(ranks).astype(np.int64)
    return mr, mrr, hits
