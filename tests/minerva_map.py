"""TEST-ONLY glue: loads a case of tests/golden/minerva_e2e.npz (tensors of the reference's PyTorch sister models in
their torch layouts + the reference outputs) and maps it with coper_amd.weights.from_minerva_state_dict."""
import numpy as np

from coper_amd import data as cdata
from coper_amd.weights import from_minerva_state_dict


def load_case(g, tag):
    E, R, B, d1, d2, C, r_dim = (int(v) for v in g[tag + ":dims"])
    sd = {k.split(":sd:")[1]: g[k] for k in g.files if k.startswith(tag + ":sd:")}
    over, p = from_minerva_state_dict(sd, g[tag + ":ent"], g[tag + ":rel"], d1, d2, cpg=(tag == "cpg"))
    md = dict(cdata._COMMON)
    md.update(over)
    q = dict(e1=g[tag + ":e1"].astype(np.int64), rel=g[tag + ":r"].astype(np.int64), e2=g[tag + ":e2"].astype(np.int64))
    return md, p, q, g[tag + ":S"], g[tag + ":S_fact"]


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))
