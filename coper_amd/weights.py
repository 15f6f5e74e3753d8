"""Weight import: parameter dicts keyed by the reference's TF leaf names from
  * an .npz / dict of arrays exported from a TF checkpoint (`tf.train.load_variable` per leaf),
  * the `best_embeddings.ckpt` pickle the reference driver writes (`run_cpg.py:242-249`):
    `[rel_emb, ent_emb]`, or `ent_emb` alone for g_lookup models,
  * the state_dict of the reference's PyTorch sister models (`CoPER_MINERVA/src/emb/fact_network.py`
    `ConvE` :116-197 and `CPG_ConvE` :261-439), whose tensors are re-laid out for the qa_cpg graph.
  * the TF `Saver` checkpoint itself (`model_weights.ckpt.index` + `.data-00000-of-00001`, `run_cpg.py:92-94,189,
    206,252`) through `coper_amd.tf_bundle` (no TensorFlow needed), both directions."""
from __future__ import annotations

import pickle
from typing import Dict

import numpy as np

__all__ = ["load_npz", "load_best_embeddings_pickle", "from_minerva_state_dict", "leaf_name", "tf_variable_name",
           "load_tf_checkpoint", "save_tf_checkpoint"]


def leaf_name(tf_variable_name: str) -> str:
    """'variables/variables/fc_weights/CPG/Projection0:0' -> 'fc_weights/CPG/Projection0' (scopes
    nest `variables/` from run_cpg.py:114 and again from models.py:169; BN layers sit one scope up)."""
    n = tf_variable_name.split(":")[0]
    while n.startswith("variables/"):
        n = n[len("variables/"):]
    return n


def tf_variable_name(leaf: str) -> str:
    """Inverse of leaf_name: BN layers are created outside the inner `variables` scope (models.py:183; the
    generators' BN at models.py:65), everything else inside it (models.py:169)."""
    if leaf.startswith(("Conv1BN/", "FCBN/")) or "/BatchNorm/" in leaf:
        return "variables/" + leaf
    return "variables/variables/" + leaf


_SLOT_SUFFIXES = ("/AMSGrad", "/AMSGrad_1", "/AMSGrad_2")     # m, v, v_hat (amsgrad.py:117-119)


def load_tf_checkpoint(prefix, with_optimizer=False):
    """`saver.save(session, prefix)` output -> parameters by leaf name (float32).  Optimizer slots
    (`<var>/AMSGrad{,_1,_2}` = m, v, v_hat, amsgrad.py:117-119) and the beta powers are dropped unless
    `with_optimizer`, which returns (params, {leaf: (m, v, v_hat)}, {"beta1_power": .., "beta2_power": ..})."""
    from . import tf_bundle
    raw = tf_bundle.read_bundle(str(prefix))
    params, slots, powers = {}, {}, {}
    for name, a in raw.items():
        leaf = leaf_name(name)
        if leaf in ("beta1_power", "beta2_power"):
            powers[leaf] = float(a)
            continue
        for k, suf in enumerate(_SLOT_SUFFIXES):
            if leaf.endswith(suf) and not leaf[:-len(suf)].endswith("/AMSGrad"):
                slots.setdefault(leaf[:-len(suf)], [None, None, None])[k] = np.asarray(a, np.float32)
                break
        else:
            if a.dtype.kind == "f":
                params[leaf] = np.asarray(a, np.float32)
    if with_optimizer:
        return params, {k: tuple(v) for k, v in slots.items()}, powers
    return params


def save_tf_checkpoint(prefix, params, slots=None, powers=None) -> None:
    """Parameters by leaf name -> a checkpoint `saver.restore(session, prefix)` of the reference graph can read
    (variable names per SURVEY.md 8-A).  `slots` {leaf: (m, v, v_hat)} and `powers` {"beta1_power", "beta2_power"}
    (ConvE.optimizer_state()) add the AMSGrad state under the names `tf.train.Saver()` expects for the full graph
    (run_cpg.py:189); without them restore with `tf.train.Saver(var_list=<model variables>)`."""
    from . import tf_bundle
    out = {tf_variable_name(k): np.asarray(v, np.float32) for k, v in params.items()}
    for leaf, parts in (slots or {}).items():
        for suf, a in zip(_SLOT_SUFFIXES, parts):
            out[tf_variable_name(leaf) + suf] = np.asarray(a, np.float32)
    for k in ("beta1_power", "beta2_power"):
        if powers and k in powers:
            out["variables/" + k] = np.float32(powers[k])
    tf_bundle.write_bundle(str(prefix), out)


def load_npz(path) -> Dict[str, np.ndarray]:
    with np.load(path) as z:
        return {leaf_name(k): np.asarray(z[k], np.float32) for k in z.files}


def load_best_embeddings_pickle(path, is_parameter_lookup=False) -> Dict[str, np.ndarray]:
    with open(path, "rb") as f:
        obj = pickle.load(f)
    if is_parameter_lookup:
        return {"ent_emb": np.asarray(obj, np.float32)}
    rel_emb, ent_emb = obj
    return {"rel_emb": np.asarray(rel_emb, np.float32), "ent_emb": np.asarray(ent_emb, np.float32)}


def from_minerva_state_dict(sd: Dict[str, np.ndarray], entity_emb, relation_emb, emb_2D_d1: int, emb_2D_d2: int,
                            cpg: bool):
    """state_dict of fact_network.ConvE / CPG_ConvE (+ the KG's embedding tables) -> (model_descriptors overrides,
    parameters by qa_cpg leaf name).  Differences folded away (fact_network.py line numbers):
      * bn0 on the input image (:150 / :356): a scalar affine in eval mode -> folded into the conv taps and bias
      * no BN after the conv (`# X = self.bn1(X)` :153 / :364)               -> Conv1BN = identity
      * NCHW flatten `X.view(-1, feat_dim)` (:156 / :368)                     -> dense-weight rows permuted to (i, j, c)
      * BatchNorm eps 1e-5 (torch default) vs 1e-3 (TF default)               -> moving_variance shifted by the difference
    The sister models apply a sigmoid to the scores (:166 / :389); this engine returns logits like qa_cpg."""
    sd = {k: np.asarray(v) for k, v in sd.items()}
    ent = np.asarray(entity_emb, np.float32)
    rel = np.asarray(relation_emb, np.float32)
    E, d = ent.shape
    R, r_dim = rel.shape
    C = sd["conv1.weight"].shape[0]
    assert sd["conv1.weight"].shape[2:] == (3, 3) and emb_2D_d1 * emb_2D_d2 == d
    md = dict(num_ent=E, num_rel=R, ent_emb_size=d, rel_emb_size=r_dim, emb_h=emb_2D_d1, emb_w=emb_2D_d2,
              conv_num_channels=C, context_rel_conv=None, context_rel_out=[] if cpg else None,
              context_rel_use_batch_norm=False, concat_rel=False)
    in_h = emb_2D_d1 if cpg else 2 * emb_2D_d1
    Ho, Wo = in_h - 2, emb_2D_d2 - 2
    F = Ho * Wo * C
    p = {"ent_emb": ent, "rel_emb": rel, "pred_bias": sd["b"].astype(np.float32)}
    a = float(sd["bn0.weight"][0] / np.sqrt(sd["bn0.running_var"][0] + np.float32(1e-5)))
    c = float(sd["bn0.bias"][0] - sd["bn0.running_mean"][0] * a)
    Wc = sd["conv1.weight"]                                                      # [C, 1, 3, 3]
    p["conv1_weights"] = (a * Wc[:, 0].transpose(1, 2, 0))[:, :, None, :].astype(np.float32)   # HWIO [3,3,1,C]
    p["conv1_bias"] = (sd["conv1.bias"] + c * Wc[:, 0].sum(axis=(1, 2))).astype(np.float32)
    p["Conv1BN/gamma"] = np.ones(C, np.float32)
    p["Conv1BN/beta"] = np.zeros(C, np.float32)
    p["Conv1BN/moving_mean"] = np.zeros(C, np.float32)
    p["Conv1BN/moving_variance"] = np.full(C, 1.0 - 1e-3, np.float32)
    pix, ch = np.divmod(np.arange(F), C)      # ours: f = (i*Wo + j)*C + ch ; reference: f_ref = ch*(Ho*Wo) + (i*Wo + j)
    f_ref = ch * (Ho * Wo) + pix
    if cpg:
        # generator networks (fact_network.py:228-259): Sequential of [Linear(no bias), (BatchNorm1d), ReLU, Dropout]* + Linear
        hidden = []
        for gname, last_perm in (("fc_weights", True), ("fc_bias", False)):
            idx = sorted({int(k.split(".")[2]) for k in sd if k.startswith(gname + ".network.")})
            lin = [i for i in idx if ("%s.network.%d.weight" % (gname, i)) in sd and sd["%s.network.%d.weight" % (gname, i)].ndim == 2]
            bns = [i for i in idx if ("%s.network.%d.running_mean" % (gname, i)) in sd]
            if any(("%s.network.%d.bias" % (gname, i)) in sd for i in lin):
                raise NotImplementedError("generator Linear layers with bias (cpg_use_bias) have no counterpart in qa_cpg (models.py:44-54)")
            for j, i in enumerate(lin):
                W = sd["%s.network.%d.weight" % (gname, i)]                       # [out, in]
                if j == len(lin) - 1 and last_perm:
                    n_in = W.shape[1]
                    P = W.T.reshape(n_in, F, d)[:, f_ref, :].reshape(n_in, F * d)
                else:
                    P = W.T
                p["%s/CPG/Projection%d" % (gname, j)] = np.ascontiguousarray(P, np.float32)
                if j < len(lin) - 1 and gname == "fc_weights":
                    hidden.append(int(W.shape[0]))
            for j, i in enumerate(bns):
                base, pre = "%s/CPG/Projection%d/BatchNorm/" % (gname, j), "%s.network.%d." % (gname, i)
                p[base + "gamma"] = sd[pre + "weight"].astype(np.float32)
                p[base + "beta"] = sd[pre + "bias"].astype(np.float32)
                p[base + "moving_mean"] = sd[pre + "running_mean"].astype(np.float32)
                p[base + "moving_variance"] = (sd[pre + "running_var"].astype(np.float64) + 1e-5 - 1e-3).astype(np.float32)
            if gname == "fc_weights":
                md["context_rel_use_batch_norm"] = bool(bns)
        md["context_rel_out"] = hidden
    else:
        p["fc_weights"] = np.ascontiguousarray(sd["fc.weight"].T[f_ref, :], np.float32)   # [F, d]
        p["fc_bias"] = sd["fc.bias"].astype(np.float32)
    p["FCBN/gamma"] = sd["bn2.weight"].astype(np.float32)
    p["FCBN/beta"] = sd["bn2.bias"].astype(np.float32)
    p["FCBN/moving_mean"] = sd["bn2.running_mean"].astype(np.float32)
    p["FCBN/moving_variance"] = (sd["bn2.running_var"].astype(np.float64) + 1e-5 - 1e-3).astype(np.float32)
    return md, p
