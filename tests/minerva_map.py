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


def load_grad_case(g, tag):
    """tests/golden/minerva_grads.npz (oracle/gen_golden.py: gen_minerva_grads): the sister model's tensors, a dense
    1-vs-all label batch, and loss + gradients from torch autograd through the reference's forward and loss."""
    E, R, B, d1, d2, C, r_dim = (int(v) for v in g[tag + ":dims"])
    sd = {k.split(":sd:")[1]: g[k] for k in g.files if k.startswith(tag + ":sd:")}
    over, p = from_minerva_state_dict(sd, g[tag + ":ent"], g[tag + ":rel"], d1, d2, cpg=(tag == "cpg"))
    md = dict(cdata._COMMON)
    md.update(over)
    md.update(hidden_dropout=0.0, output_dropout=0.0, input_dropout=0.0, context_rel_dropout=0.0, batch_norm_train_stats=False,
              label_smoothing_epsilon=float(g[tag + ":eps_ls"]), learning_rate=1e-3, use_negative_sampling=False)
    batch = dict(e1=g[tag + ":e1"].astype(np.int64), rel=g[tag + ":r"].astype(np.int64), labels=g[tag + ":labels"].astype(np.float32))
    return md, p, sd, batch


def reference_grads_in_our_layout(g, tag, sd, ours):
    """(expected, got) pairs per leaf.  Leaves whose layout is a pure re-indexing of a sister tensor are expected =
    re-indexed sister gradient, got = ours; for the conv filters (input BN folded in: K' = a K, kb' = kb + c sum K) the
    comparison runs on the sister's side by the chain rule: dL/dK = a dL/dK' + c dL/dkb', dL/dkb = dL/dkb'."""
    E, R, B, d1, d2, C, r_dim = (int(v) for v in g[tag + ":dims"])
    d = d1 * d2
    cpg = tag == "cpg"
    in_h = d1 if cpg else 2 * d1
    Ho, Wo = in_h - 2, d2 - 2
    F = Ho * Wo * C
    pix, ch = np.divmod(np.arange(F), C)
    f_ref = ch * (Ho * Wo) + pix
    G = lambda k: g[tag + ":grad:" + k]
    pairs = {"ent_emb": (G("ent"), ours["ent_emb"]), "rel_emb": (G("rel"), ours["rel_emb"]), "pred_bias": (G("b"), ours["pred_bias"]),
             "FCBN/gamma": (G("bn2.weight"), ours["FCBN/gamma"]), "FCBN/beta": (G("bn2.bias"), ours["FCBN/beta"])}
    if cpg:
        gw = G("fc_weights.network.0.weight")                                  # [F_ref * d, r]
        pairs["fc_weights/CPG/Projection0"] = (gw.T.reshape(r_dim, F, d)[:, f_ref, :].reshape(r_dim, F * d),
                                               ours["fc_weights/CPG/Projection0"])
        pairs["fc_bias/CPG/Projection0"] = (G("fc_bias.network.0.weight").T, ours["fc_bias/CPG/Projection0"])
    else:
        pairs["fc_weights"] = (G("fc.weight").T[f_ref, :], ours["fc_weights"])
        pairs["fc_bias"] = (G("fc.bias"), ours["fc_bias"])
    a = float(sd["bn0.weight"][0] / np.sqrt(sd["bn0.running_var"][0] + np.float32(1e-5)))
    c = float(sd["bn0.bias"][0] - sd["bn0.running_mean"][0] * a)
    dK = np.asarray(ours["conv1_weights"], np.float64).reshape(3, 3, C)       # HWIO with I = 1
    dkb = np.asarray(ours["conv1_bias"], np.float64).reshape(C)
    pairs["conv1.weight"] = (G("conv1.weight")[:, 0], a * dK.transpose(2, 0, 1) + c * dkb[:, None, None])
    pairs["conv1.bias"] = (G("conv1.bias"), dkb)
    return pairs
