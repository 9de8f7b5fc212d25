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
        self._drop_done: Optional[torch.Tensor] = None

    def _keep_bits(self, x: torch.Tensor):
        """Keep bits of Dropout(0.5) on the patch features (ABMIL.py:49: the dropped x is also what gets pooled).  No dropped
        copy: a Philox keep-bit tensor (1/32 of x) that the gate, pool and backward kernels read x through
        (csrc/dropout.hip).  seed: torch's generator, so torch.manual_seed governs the masks; offset: a per-module pass
        counter kept ON THE DEVICE, so a step replayed from a hipGraph (graph_step.py) draws a fresh mask every replay."""
        self._drop_state(x.device)
        xbits, _ = ops.dropout_keep_bits_pair(x.shape[0], 0, x.shape[1], self._drop_seed, 0, self._drop_ctr, self._drop_done)
        return xbits

    def _drop_state(self, device):
        if self._drop_seed is None:
            self._drop_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        if self._drop_ctr is None or self._drop_ctr.device != device:
            self._drop_ctr = torch.zeros(1, device=device, dtype=torch.int32)
            self._drop_done = torch.zeros(1, device=device, dtype=torch.int32)     # sign-off word of the generator launch

    def _gate_params(self):
        return (self.attention_V[0].weight, self.attention_V[0].bias, self.attention_U[0].weight, self.attention_U[0].bias,
                self.attention_weights.weight, self.attention_weights.bias)

    def flat(self, x: torch.Tensor, layout: BagLayout) -> torch.Tensor:
        """x [R, L] rows of all bags -> M [B, L]."""
        xbits = None
        if self.training and x.shape[1] % 32 == 0:
            xbits = self._keep_bits(x)
        elif self.training:
            x = F.dropout(x, 0.5, True)
        self.last_xbits = xbits
        M, scores = ops.gated_attention_pool(x, *self._gate_params(), layout, xbits)
        self.last_scores = scores
        return M

    FUSED_HEAD_L = (256, 512, 768, 1024)

    def can_fuse_head(self, x: torch.Tensor, num_classes: int) -> bool:
        return x.shape[1] in self.FUSED_HEAD_L and num_classes <= 4

    def flat_head_loss(self, x: torch.Tensor, layout: BagLayout, fc_weight, fc_bias, y, scale: float, loss_kind: int = 0,
                       head_train: bool = False):
        """Pool + Dropout(.25) + head + sigmoid + loss in one autograd node (ops.gated_pool_head_loss): rows x [R, L] of all
        bags, labels y [B, C] -> (loss, prob, logits).  Used by aggregator.forward(labels=...)."""
        xbits = mbits = None
        if self.training or head_train:
            # patch keep bits (Dropout(.5), ABMIL.py:49) and the head's keep words (Dropout(.25), aggregator.py:129: inside
            # the fused tail) from one generator launch (the words one stream position behind the bits, where the separate launch drew them), which also advances the pass counter
            self._drop_state(x.device)
            xbits, mbits = ops.dropout_keep_bits_pair(x.shape[0] if self.training else 0, layout.B if head_train else 0,
                                                      x.shape[1], self._drop_seed, self._drop_seed ^ 0x9E3779B97F4A7C15,
                                                      self._drop_ctr, self._drop_done)
        self.last_xbits, self.last_mbits = xbits, mbits
        loss, prob, z, M = ops.gated_pool_head_loss(x, *self._gate_params(), fc_weight, fc_bias, y, layout, scale, loss_kind,
                                                    xbits, mbits)
        self.last_M = M
        return loss, prob, z

    def forward(self, x: torch.Tensor, lengths: Optional[Sequence[int]] = None) -> torch.Tensor:
        if x.dim() == 2:
            x = x.unsqueeze(0)
        B, N, L = x.shape
        layout = BagLayout.make(lengths, x.device) if lengths is not None else BagLayout.uniform(B, N, x.device)
        return self.flat(x.reshape(B * N, L), layout)          # [B, L]; [1, L] for the reference's one-bag call
