"""Shared by the GPU tests: the logits that DEFINE a mode's ranks.

f32 mode: the mode's own logits (they are the documented fp32 chain, bit for bit).  bf16x3 mode: its count kernel decides
only comparisons wider than its own error; closer ones go to the fp32 chain (the exact band), so its ranks and tie counts
are those of the C restatement of that chain applied to the mode's h -- the mode's own logits (score_all, top-k values)
stay within the 1e-3 gate of it but do not define the ranks."""
import numpy as np


def rank_defining_logits(O, m, h, params):
    if m.score_mode == "f32":
        return m.score_all(h).cpu().numpy()
    lo, hi = m.shard
    return O.score_chain(np.ascontiguousarray(h.cpu().numpy()), np.ascontiguousarray(np.asarray(params["ent_emb"], np.float32)[lo:hi]),
                         np.ascontiguousarray(np.asarray(params["pred_bias"], np.float32)[lo:hi]))
