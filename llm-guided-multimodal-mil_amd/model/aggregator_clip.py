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
        h = M
        if self.training and M.shape[1] % 32 == 0:
            # Dropout(.25) in front of the head (aggregator_clip.py / aggregator.py:129) from the SAME stream as the fused
            # route: Philox keep words keyed like ABMIL.flat_head_loss's (module seed ^ golden ratio, the pass counter the
            # patch mask of this forward has just advanced), not torch's generator (two training-mode streams)
            ab = self.extractor_pathology
            self.last_mbits = ops.dropout_keep_bits(M.shape[0], M.shape[1], ops.M_DROP_P, ab._drop_seed ^ 0x9E3779B97F4A7C15,
                                                    0, M.device, offset_dev=ab._drop_ctr)
            h = ops.dropout_bits(M, self.last_mbits, ops.M_DROP_SCALE)
        elif self.training:
            h = F.dropout(M, 0.25, True)
        p, z = ops.head_sigmoid(h, self.fc[1].weight, self.fc[1].bias)
        self.last_logits = z
        return M, p
