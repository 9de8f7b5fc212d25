"""Image-only variant (reference: model/aggregator_clip.py:6-118, pathology branch :109-118):
`forward(x_list) -> (bag_embedding [B, L], sigmoid(fc(bag_embedding)) [B, C])` with ABMIL straight on
the patch features.  This is BASELINE config 2's model; `trainer.ImageOnlyTrainer` is its fused step."""
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .dim1 import ABMIL


class aggregator(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        if list(args.modality) != ["pathology"] or getattr(args, "model_pathology", "ABMIL") != "ABMIL":
            raise NotImplementedError("aggregator_clip: only modality ['pathology'] with ABMIL is on the built path")
        L = int(getattr(args, "patch_dim", 768))                     # aggregator_clip.py:36
        self.extractor_pathology = ABMIL(args, L=L)
        self.fc = nn.Sequential(nn.Dropout(0.25), nn.Linear(L, args.num_classes))
        self.last_logits: Optional[torch.Tensor] = None

    def forward(self, x_list: List[torch.Tensor], lengths=None):
        M = self.extractor_pathology(x_list[0], lengths)
        h = F.dropout(M, 0.25, True) if self.training else M
        p, z = ops.head_sigmoid(h, self.fc[1].weight, self.fc[1].bias)
        self.last_logits = z
        return M, p
