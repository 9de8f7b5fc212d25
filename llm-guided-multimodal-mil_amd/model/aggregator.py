"""`aggregator(args)`: the drop-in boundary of the path (reference: model/aggregator.py:9-209).

Same constructor (`args.modality`, `args.model_pathology`, `args.model_CI`, `args.aggregator`,
`args.num_classes`), same `forward(x_list, x_CI)` and return tuples, same state_dict keys - but every
tensor op on the path runs through the HIP library:

    fc_pathology (GEMM + tanh) -> clinic_extractor (frozen CLIP text tower, cached per note)
    -> fc_CI2Pth (GEMM + tanh) -> TwoWayTransformer (text <-> patches) -> multi-modal bag [text | patches]
    -> aggregator (gated-attention MIL pool) -> fc + sigmoid.

Differences, all deliberate: a batch is B independent bags (the reference's B>1 ABMIL
degenerates to a sum); CT encoders / TransMIL / tabular-CI MLPs are out of scope and raise; the positional
table lives on the device (the reference re-uploads `pe[:, :N]` every forward, aggregator.py:160-190);
`last_logits` exposes the pre-sigmoid scores for the parity checks."""
import math
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import lifetime, ops
from ..bags import BagLayout
from .sam.transformer import TwoWayTransformer

EMBED = 512


def _arg(args, name, default):
    return getattr(args, name, default)


class aggregator(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        modality = list(args.modality)
        if "CT" in modality:
            raise NotImplementedError("CT encoders (torchvision/MONAI 3-D backbones) are outside the MIL hot path "
                                      "(SURVEY.md section 2 #14); use modality ['pathology'] or ['CI']")
        mk_twoway = lambda: TwoWayTransformer(args=args, depth=2, embedding_dim=EMBED, num_heads=8, mlp_dim=2048)  # noqa: E731
        self.fc_CI2CT = nn.Sequential(nn.Linear(EMBED, EMBED), nn.Tanh())                 # aggregator.py:44
        if "pathology" in modality:
            self.fc_pathology = nn.Sequential(nn.Linear(768, EMBED), nn.Tanh())           # :47
            if _arg(args, "model_pathology", "ABMIL") == "ABMIL":
                from .dim1 import ABMIL
                self.extractor_pathology = ABMIL(args, L=EMBED)                           # :50 (built, never called)
            elif args.model_pathology not in ("-", None):
                raise NotImplementedError(f"model_pathology={args.model_pathology}: only ABMIL is on the built path")
            self.TwoWayTransformer_Pth = mk_twoway()                                      # :58
        self.fc_CI2Pth = nn.Sequential(nn.Linear(EMBED, EMBED), nn.Tanh())                # :66
        self.fc_CI = nn.Sequential(nn.Linear(EMBED, EMBED), nn.Tanh())                    # :68
        self.TwoWayTransformer_Both = mk_twoway()                                         # :70
        agg = _arg(args, "aggregator", "ABMIL")
        if agg == "ABMIL":
            from .dim1 import ABMIL
            self.aggregator = ABMIL(args, L=EMBED)                                        # :81
        elif agg != "-":
            raise NotImplementedError(f"aggregator={agg}: TransMIL needs the absent nystrom_attention package "
                                      "(parity unpinned, SURVEY.md section 8f); ABMIL_v2 is not on the built path")
        if _arg(args, "model_CI", "CLIP") == "CLIP":
            from .dim1 import CLIP
            self.clinic_extractor = CLIP(args)                                            # :122
        else:
            raise NotImplementedError(f"model_CI={args.model_CI}: only the CLIP text extractor is on the built path")
        self.prompt_embedding = nn.Parameter(torch.randn(1, EMBED))                       # :124 (unused upstream too)
        self.fc = nn.Sequential(nn.Dropout(0.25), nn.Linear(EMBED, args.num_classes))     # :128-131
        self._pe: Optional[torch.Tensor] = None
        self.last_logits: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------ positional table (aggregator.py:99-106)
    def pe_rows(self, n: int, device) -> torch.Tensor:
        """First n rows of the sinusoidal table, device-resident, grown on demand.  Computed on the host with
        the reference's own expressions (bit-identical to `self.pe[:, :n]`) and uploaded once."""
        if self._pe is None or self._pe.shape[0] < n or self._pe.device != device:
            rows = max(n, 2048, 0 if self._pe is None else 2 * self._pe.shape[0])
            pe = torch.zeros((rows, EMBED))
            position = torch.arange(0, rows).unsqueeze(1)
            div_term = torch.exp(torch.arange(0, EMBED, 2, dtype=torch.float) * -(math.log(10000.0) / EMBED))
            pe[:, 0::2] = torch.sin(position.float() * div_term)
            pe[:, 1::2] = torch.cos(position.float() * div_term)
            self._pe = pe.to(device)
        return lifetime.note(self._pe)            # a captured graph keeps the table it saw when a larger one replaces it

    def _lin_tanh(self, seq: nn.Sequential, x):
        return ops.linear_act(x, seq[0].weight, seq[0].bias, "tanh")

    def _head(self, M):
        if self.training:
            M = F.dropout(M, 0.25, True)
        p, z = ops.head_sigmoid(M, self.fc[1].weight, self.fc[1].bias)
        self.last_logits = z
        return p

    # ------------------------------------------------------------------ forward (aggregator.py:134-209)
    def forward(self, x_list: List[torch.Tensor], x_CI: torch.Tensor, lengths: Optional[List[int]] = None,
                text_features: Optional[torch.Tensor] = None):
        """x_list = [x_pathology [B, N, 768]] (or [] for CI only); x_CI int64 [B, P, ctx] token ids.
        `lengths` (optional) gives the true patch count of each zero-padded bag (dataset.py:386-391 pads to a
        fixed length when batch > 1); padded rows are then dropped instead of being attended to."""
        modality = self.args.modality
        # text_features [B, P, 512] (optional): embeddings of the frozen text tower computed earlier by
        # `self.clinic_extractor(x_CI)`; lets a captured hipGraph replay the trainable part only
        t = text_features if text_features is not None else self.clinic_extractor(x_CI)   # :151  [B, P, 512]
        B, P, _ = t.shape
        if "pathology" in modality:
            x = x_list[0]
            if x.dim() == 2:
                x = x.unsqueeze(0)
            N = x.shape[1]
            if lengths is None:
                n_len = [N] * B
                flat = x.reshape(B * N, x.shape[2])
            else:
                n_len = [int(v) for v in lengths]
                flat = torch.cat([x[b, :n] for b, n in enumerate(n_len)], 0)
            xi = self._lin_tanh(self.fc_pathology, flat)                                  # :149
            point = self._lin_tanh(self.fc_CI2Pth, t.reshape(B * P, EMBED))               # :190
            q, k = self.TwoWayTransformer_Pth.flat(xi, point, self.pe_rows(max(n_len), xi.device), n_len, [P] * B,
                                                   keys_tail_rows=B * P)
            # multi-modal bag per patient = its text tokens + its patch tokens (:192).  Rows are kept as
            # [all patches | all tokens] (the patch tokens stay where the last LayerNorm wrote them, only the few
            # token rows are appended) and the tile map tells the pool kernels which rows belong to which bag;
            # attention pooling does not depend on the order of a bag's rows.
            x0 = ops.append_rows(k, q, tail_reserved=True)     # k comes from the last block's norm4 with keys_tail_rows
            layout = BagLayout.two_segment(n_len, [P] * B, x0.device)
            M = self.aggregator.flat(x0, layout) if hasattr(self, "aggregator") else x0   # :198-199
            return self._head(M), q.view(B, P, EMBED)                                     # :200,207
        if "CI" in modality:
            x0 = self._lin_tanh(self.fc_CI, t.reshape(B * P, EMBED))                      # :195
            M = self.aggregator.flat(x0, BagLayout.uniform(B, P, x0.device)) if hasattr(self, "aggregator") else x0
            return self._head(M)                                                          # :209
        raise NotImplementedError(f"modality {modality}")
