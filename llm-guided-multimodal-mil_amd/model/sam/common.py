"""MLP block of the two-way transformer (reference: model/sam/common.py:13-26): lin2(act(lin1(x))).
Both linears run on the fp32-MFMA GEMM; the activation (ReLU, chosen at sam/transformer.py:18) and the
caller's residual are fused into the GEMM epilogues."""
import torch.nn as nn

from ... import ops


class MLPBlock(nn.Module):
    def __init__(self, embedding_dim: int, mlp_dim: int, act: str = "relu"):
        super().__init__()
        self.lin1 = nn.Linear(embedding_dim, mlp_dim)
        self.lin2 = nn.Linear(mlp_dim, embedding_dim)
        self.act = act

    def forward(self, x, residual=None):
        h = ops.linear_act(x, self.lin1.weight, self.lin1.bias, self.act)
        return ops.linear_act(h, self.lin2.weight, self.lin2.bias, "none", residual=residual)
