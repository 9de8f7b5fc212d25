"""Gated-attention MIL aggregator (reference: model/dim1/ABMIL.py:6-64), MI355X path.

Same constructor, parameter names (attention_V.0, attention_U.0, attention_weights) and B=1 semantics.
A batch [B, N, L] is treated as B independent bags (one softmax per bag) - the reference's B>1 case
degenerates to a plain sum over N (ABMIL.py:48,57) and is deliberately NOT reproduced.  D is fixed at 192."""
from typing import Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...bags import BagLayout


class ABMIL(nn.Module):
    def __init__(self, args=None, L: int = 768, D: int = 192, K: int = 1):
        super().__init__()
        if D != ops.GATE_D or K != 1:
            raise NotImplementedError("the HIP gate kernels are built for D=192, K=1 (the reference never changes them)")
        self.L, self.D, self.K = L, D, K
        self.attention_V = nn.Sequential(nn.Linear(L, D), nn.Tanh())
        self.attention_U = nn.Sequential(nn.Linear(L, D), nn.Sigmoid())
        self.attention_weights = nn.Linear(D, K)
        self.dropout1 = nn.Dropout(0.5)
        self.last_scores: Optional[torch.Tensor] = None
        self.last_xbits: Optional[torch.Tensor] = None
        self._drop_seed: Optional[int] = None
        self._drop_ctr: Optional[torch.Tensor] = None

    def flat(self, x: torch.Tensor, layout: BagLayout) -> torch.Tensor:
        """x [R, L] rows of all bags -> M [B, L]."""
        xbits = None
        if self.training and x.shape[1] % 32 == 0:
            # ABMIL.py:49: the dropped x is also what gets pooled.  No dropped copy: a Philox keep-bit tensor (1/32 of x)
            # that the gate, pool and backward kernels read x through (csrc/dropout.hip).  seed: torch's generator, so
            # torch.manual_seed governs the masks; offset: a per-module pass counter kept ON THE DEVICE, so a step replayed
            # from a hipGraph (graph_step.py) draws a fresh mask every replay.
            if self._drop_seed is None or self._drop_ctr is None or self._drop_ctr.device != x.device:
                self._drop_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
                self._drop_ctr = torch.zeros(1, device=x.device, dtype=torch.int32)
            xbits = ops.dropout_keep_bits(x.shape[0], x.shape[1], ops.X_DROP_P, self._drop_seed, 0, x.device,
                                          offset_dev=self._drop_ctr)
            ops.counter_add(self._drop_ctr, 1)
        elif self.training:
            x = F.dropout(x, 0.5, True)
        self.last_xbits = xbits
        M, scores = ops.gated_attention_pool(x, self.attention_V[0].weight, self.attention_V[0].bias,
                                             self.attention_U[0].weight, self.attention_U[0].bias,
                                             self.attention_weights.weight, self.attention_weights.bias, layout, xbits)
        self.last_scores = scores
        return M

    def forward(self, x: torch.Tensor, lengths: Optional[Sequence[int]] = None) -> torch.Tensor:
        if x.dim() == 2:
            x = x.unsqueeze(0)
        B, N, L = x.shape
        layout = BagLayout.make(lengths, x.device) if lengths is not None else BagLayout.uniform(B, N, x.device)
        return self.flat(x.reshape(B * N, L), layout)          # [B, L]; [1, L] for the reference's one-bag call
