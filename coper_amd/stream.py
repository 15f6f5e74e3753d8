"""A stream of evaluation passes with everything but the kernels of a pass off its critical path.

What `ranking_and_hits` does per `session.run` (metrics.py:38-57) -- one batch of (e1, rel, e2, known answers) scored against
all entities and ranked -- as a pipeline over one HIP stream:

    pass n runs;  beside its encoder launch  * 16 extra workgroups read batch n + 1 over PCIe      (coper_stage_ids_next)
                                             * one more sorts batch n + 1 by relation               (coper_group_next)
                  beside pass n + 1's first launch the ranks of pass n leave for pinned host memory (coper_post_i32_next)

No copy engine, no launch that only moves data except in front of the first pass and behind the last one.  `bench.py --h2d overlap`
times exactly this pattern (with its own loop: it measures); `examples/eval_stream.py` and the tests use this class.

THE GUARD (include/coper_hip.h: coper_group_next).  Pass n + 1 runs on a sorting made while pass n ran; if the ids at the staged
addresses change in between, the library notices on the device and writes COPER_RANK_STALE into every rank of that pass.  `run`
checks every pass's ranks when it collects them and ranks such a batch again with a plain call: a stale grouping costs a pass,
never a wrong rank."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

__all__ = ["RankStream"]

KEYS = ("e1", "rel", "e2", "filt_indptr", "filt_idx")


class RankStream(object):
    def __init__(self, model, max_queries: int, max_filter_nnz: int):
        self.model = model
        self.dev = model.device
        self.Q, self.nnz = int(max_queries), int(max_filter_nnz)
        width = 4 * self.Q + 1 + self.nnz
        # two staging arrays (the pass that runs reads one while the next batch arrives in the other), two rank buffers
        self.stages = [torch.empty(width, dtype=torch.int64, device=self.dev) for _ in range(2)]
        self.ranks_dev = [torch.empty(max(1, self.Q), dtype=torch.int32, device=self.dev) for _ in range(2)]
        model.reserve(self.Q, self.nnz)
        self.stale_passes = 0          # passes of all runs whose prepared grouping was found stale (each was ranked again)

    def pack(self, batch: dict):
        """[e1 | rel | e2 | filt_indptr | filt_idx] as int32 in ONE pinned buffer, and the offsets of the five arrays."""
        arrs = [np.ascontiguousarray(np.asarray(batch[k])).reshape(-1) for k in KEYS]
        B, nnz = len(arrs[0]), len(arrs[4])
        if B > self.Q or nnz > self.nnz:
            raise ValueError("RankStream: a batch of %d queries / %d filter entries exceeds the stream's capacity (%d / %d)" % (B, nnz, self.Q, self.nnz))
        if len(arrs[1]) != B or len(arrs[2]) != B or len(arrs[3]) != B + 1:
            raise ValueError("RankStream: e1, rel, e2 of one length B and filt_indptr of B + 1")
        for a in arrs:
            if a.dtype.kind not in "iu" or (a.size and (int(a.max()) >= 2 ** 31 or int(a.min()) < -2 ** 31)):
                raise ValueError("RankStream: ids must be integers that fit int32")
        sizes = [int(a.size) for a in arrs]
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        pin = torch.empty(max(1, int(offs[-1])), dtype=torch.int32).pin_memory()
        host = pin.numpy()
        for a, o in zip(arrs, offs):
            host[o:o + a.size] = a
        # (and pinned memory for the ranks of this batch: allocated once per packed batch, not per run)
        return dict(pin=pin[:int(offs[-1])], offs=offs, sizes=sizes, B=B, nnz=nnz, host=torch.empty(max(1, B), dtype=torch.int32).pin_memory())

    def _views(self, c: int, pk: dict):
        st = self.stages[c]
        return [st[o:o + n] for o, n in zip(pk["offs"], pk["sizes"])]

    def run(self, batches: Sequence[dict], group_ahead: bool = True, on_pass: Optional[Callable[[int], None]] = None) -> List[np.ndarray]:
        """Ranks (int32 ndarray [B_n]) of every batch.  `batches`: dicts of the five arrays, or what `pack` returned (a host that
        scores the same batches again packs them once).  `on_pass(n)`: called after pass n was queued (progress; tests use it to
        disturb the stream)."""
        m = self.model
        packed = [b if "pin" in b else self.pack(b) for b in batches]
        hosts = [pk["host"] for pk in packed]
        live = [n for n, pk in enumerate(packed) if pk["B"] > 0]
        first = True
        for j, n in enumerate(live):
            pk, c = packed[n], j & 1
            if first:                                                      # the first batch has no pass to arrive beside
                m.widen_ids(pk["pin"], out=self.stages[c][:pk["pin"].numel()])
                first = False
            if j + 1 < len(live):
                nx = packed[live[j + 1]]
                m.stage_next(nx["pin"], self.stages[1 - c][:nx["pin"].numel()])      # batch n + 1: read beside this pass's encoder launch
                if group_ahead:
                    v1 = self._views(1 - c, nx)
                    m.group_next(v1[0], v1[1])                                       # ... and sorted by relation there
            v = self._views(c, pk)
            r, _ = m.rank_pass(v[0], v[1], v[2], v[3], v[4], filt_nnz=pk["nnz"], want_equal=False, out=self.ranks_dev[c][:pk["B"]])
            m.post_next(r, hosts[n][:pk["B"]])                                       # ranks n: out beside pass n + 1's first launch
            if on_pass is not None:
                on_pass(n)
        m.post_flush()                                                               # ... and the last ones by a launch of their own
        torch.cuda.current_stream(self.dev).synchronize()
        out = []
        for n, pk in enumerate(packed):
            r = hosts[n][:pk["B"]].numpy()
            if pk["B"] and int(r.min()) < 1:                                         # the guard fired: rank this batch again, plainly
                if int(r.max()) >= 1:          # (a stale pass has EVERY rank at COPER_RANK_STALE + its counts: all negative)
                    raise RuntimeError("RankStream: pass %d returned a mixture of ranks and values below 1" % n)
                self.stale_passes += 1
                host = pk["pin"].numpy()
                arrs = [host[o:o + s].astype(np.int64) for o, s in zip(pk["offs"], pk["sizes"])]
                r = m.rank_pass(arrs[0], arrs[1], arrs[2], arrs[3], arrs[4], want_equal=False)[0].cpu().numpy()
            out.append(r.copy())
        if self.stale_passes and hasattr(m, "stale_passes"):
            m.stale_passes()                                                         # (read and reset the library's counter)
        return out
