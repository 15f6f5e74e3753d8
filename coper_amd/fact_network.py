"""Inference entry points of the reference's PyTorch sister models on the HIP engine:
`CoPER_MINERVA/src/emb/fact_network.py` `ConvE.forward / forward_fact` (:139-197) and `CPG_ConvE.forward /
forward_fact` (:339-439) -- the scorers `EmbeddingBasedMethod.predict / forward_fact` (`src/emb/emb.py:32-68`) and
the reward shaping of `rs_pg.py:63-95` call.  Same argument order and return shapes; scores are sigmoids of the
engine's logits (`fact_network.py:166,389,437`).  The `kg` argument of the reference methods only supplies the
embedding tables, which this class takes at construction."""
from __future__ import annotations

import numpy as np
import torch

from . import data as cdata
from .models import ConvE
from .weights import from_minerva_state_dict

__all__ = ["FactNetworkScorer"]


class FactNetworkScorer(object):
    def __init__(self, state_dict, entity_embeddings, relation_embeddings, emb_2D_d1, emb_2D_d2, cpg, device=None,
                 score_mode="bf16x3"):
        """state_dict: of fact_network.ConvE (cpg=False) or CPG_ConvE with cpg_fc_net=[] (cpg=True), tensors or
        arrays; entity / relation embeddings: the KG's tables (`kg.get_all_entity_embeddings()` etc.)."""
        sd = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in state_dict.items()}
        ent = entity_embeddings.detach().cpu().numpy() if hasattr(entity_embeddings, "detach") else np.asarray(entity_embeddings)
        rel = relation_embeddings.detach().cpu().numpy() if hasattr(relation_embeddings, "detach") else np.asarray(relation_embeddings)
        over, params = from_minerva_state_dict(sd, ent, rel, int(emb_2D_d1), int(emb_2D_d2), cpg=bool(cpg))
        md = dict(cdata._COMMON)
        md.update(over)
        self.model = ConvE(md, device=device, score_mode=score_mode).load_parameters(params).prepare()

    def forward(self, e1, r, kg=None):
        """[B, num_entities] scores of every tail (`fact_network.py:139-167 / 339-391`)."""
        return torch.sigmoid(self.model.score_all(self.model.encode(e1, r)))

    def forward_fact(self, e1, r, e2, kg=None):
        """[B, 1] scores of the given facts (`fact_network.py:169-197 / 393-439`)."""
        h = self.model.encode(e1, r)
        e2 = np.asarray(e2.detach().cpu() if hasattr(e2, "detach") else e2).reshape(-1, 1).astype(np.int32)
        return torch.sigmoid(self.model.score_lookup(h, e2))

    def close(self):
        self.model.close()
