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
    over, p = from_minerva_state_dict(sd, g[tag + ":ent"], g[tag + ":rel"], d1, d2, cpg=tag.startswith("cpg"))
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
    cpg = tag.startswith("cpg")
    in_h = d1 if cpg else 2 * d1
    Ho, Wo = in_h - 2, d2 - 2
    F = Ho * Wo * C
    pix, ch = np.divmod(np.arange(F), C)
    f_ref = ch * (Ho * Wo) + pix
    G = lambda k: g[tag + ":grad:" + k]
    pairs = {"ent_emb": (G("ent"), ours["ent_emb"]), "rel_emb": (G("rel"), ours["rel_emb"]), "pred_bias": (G("b"), ours["pred_bias"]),
             "FCBN/gamma": (G("bn2.weight"), ours["FCBN/gamma"]), "FCBN/beta": (G("bn2.bias"), ours["FCBN/beta"])}
    if cpg:
        # generator networks: Linear layers in order -> Projection<j> (the last one of fc_weights re-indexed NCHW -> NHWC),
        # BatchNorm1d layers in order -> Projection<j>/BatchNorm
        for gname in ("fc_weights", "fc_bias"):
            idx = sorted({int(k.split(".")[2]) for k in sd if k.startswith(gname + ".network.")})
            lin = [i for i in idx if sd.get("%s.network.%d.weight" % (gname, i), np.zeros(1)).ndim == 2]
            bns = [i for i in idx if ("%s.network.%d.running_mean" % (gname, i)) in sd]
            for j, i in enumerate(lin):
                gw = G("%s.network.%d.weight" % (gname, i))                   # [out, in]
                if j == len(lin) - 1 and gname == "fc_weights":
                    n_in = gw.shape[1]
                    want = gw.T.reshape(n_in, F, d)[:, f_ref, :].reshape(n_in, F * d)
                else:
                    want = gw.T
                pairs["%s/CPG/Projection%d" % (gname, j)] = (want, ours["%s/CPG/Projection%d" % (gname, j)])
            for j, i in enumerate(bns):
                base = "%s/CPG/Projection%d/BatchNorm/" % (gname, j)
                pairs[base + "gamma"] = (G("%s.network.%d.weight" % (gname, i)), ours[base + "gamma"])
                pairs[base + "beta"] = (G("%s.network.%d.bias" % (gname, i)), ours[base + "beta"])
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
