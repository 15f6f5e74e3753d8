"""TEST-ONLY: maps the tensors of the reference's PyTorch sister models (CoPER_MINERVA/src/emb/fact_network.py,
stored in tests/golden/minerva_e2e.npz with their torch layouts) onto the qa_cpg leaf names / layouts, so that
the reference's own `forward` output pins conv -> dense -> score end to end.

Differences folded away here (fact_network.py line numbers):
  * bn0 on the input image (:150 / :356), a scalar affine a*x + c in eval mode -> folded into the conv taps and bias
  * no BN after the conv (`# X = self.bn1(X)` :153 / :364)            -> Conv1BN set to the identity
  * NCHW flatten `X.view(-1, feat_dim)` (:156 / :368)                  -> rows of the dense weights permuted to (i, j, c)
  * BatchNorm eps 1e-5 (torch default) vs 1e-3 (TF default)            -> moving_variance shifted by the difference
  * sigmoid on the scores (:166 / :389)                                -> compared after sigmoid
"""
import numpy as np

from coper_amd import data as cdata


def load_case(g, tag):
    E, R, B, d1, d2, C, r_dim = (int(v) for v in g[tag + ":dims"])
    d = d1 * d2
    sd = {k.split(":sd:")[1]: g[k] for k in g.files if k.startswith(tag + ":sd:")}
    cpg = tag == "cpg"
    md = dict(cdata._COMMON, num_ent=E, num_rel=R, ent_emb_size=d, rel_emb_size=r_dim, emb_h=d1, emb_w=d2,
              conv_num_channels=C, context_rel_conv=None, context_rel_out=[] if cpg else None,
              context_rel_use_batch_norm=False)
    in_h = d1 if cpg else 2 * d1
    Ho, Wo = in_h - 2, d2 - 2
    F = Ho * Wo * C
    p = {"ent_emb": g[tag + ":ent"], "rel_emb": g[tag + ":rel"], "pred_bias": sd["b"]}
    # bn0 (eval): a*x + c
    a = float(sd["bn0.weight"][0] / np.sqrt(sd["bn0.running_var"][0] + np.float32(1e-5)))
    c = float(sd["bn0.bias"][0] - sd["bn0.running_mean"][0] * a)
    Wc = sd["conv1.weight"]                                        # [C, 1, 3, 3]
    p["conv1_weights"] = (a * Wc[:, 0].transpose(1, 2, 0))[:, :, None, :].astype(np.float32)   # [3,3,1,C]
    p["conv1_bias"] = (sd["conv1.bias"] + c * Wc[:, 0].sum(axis=(1, 2))).astype(np.float32)
    p["Conv1BN/gamma"] = np.ones(C, np.float32)
    p["Conv1BN/beta"] = np.zeros(C, np.float32)
    p["Conv1BN/moving_mean"] = np.zeros(C, np.float32)
    p["Conv1BN/moving_variance"] = np.full(C, 1.0 - 1e-3, np.float32)
    # our flatten index f = (i*Wo + j)*C + ch  <->  reference f_ref = ch*(Ho*Wo) + (i*Wo + j)
    pix, ch = np.divmod(np.arange(F), C)
    f_ref = ch * (Ho * Wo) + pix
    if cpg:
        Wg = sd["fc_weights.network.0.weight"]                      # [F*d, r]
        P = Wg.T.reshape(r_dim, F, d)[:, f_ref, :].reshape(r_dim, F * d)
        p["fc_weights/CPG/Projection0"] = np.ascontiguousarray(P, np.float32)
        p["fc_bias/CPG/Projection0"] = np.ascontiguousarray(sd["fc_bias.network.0.weight"].T, np.float32)
    else:
        p["fc_weights"] = np.ascontiguousarray(sd["fc.weight"].T[f_ref, :], np.float32)   # [F, d]
        p["fc_bias"] = sd["fc.bias"].astype(np.float32)
    p["FCBN/gamma"] = sd["bn2.weight"]
    p["FCBN/beta"] = sd["bn2.bias"]
    p["FCBN/moving_mean"] = sd["bn2.running_mean"]
    p["FCBN/moving_variance"] = (sd["bn2.running_var"].astype(np.float64) + 1e-5 - 1e-3).astype(np.float32)
    q = dict(e1=g[tag + ":e1"].astype(np.int64), rel=g[tag + ":r"].astype(np.int64), e2=g[tag + ":e2"].astype(np.int64))
    return md, p, q, g[tag + ":S"], g[tag + ":S_fact"]


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))
