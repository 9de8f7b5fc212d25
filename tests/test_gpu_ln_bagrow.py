"""GPU: LayerNorm(x + o[bag of row]) as one node (one-text-token image->token attention + norm4,
sam/transformer.py:303-309) against torch's LayerNorm on the materialised sum - outputs and all four gradients."""
import pytest
import torch

from conftest import rel_err
from mil_amd import ops
from mil_amd.segments import AttnSegs

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("lengths,E", [([300, 77, 512, 129], 512), ([1024] * 8, 512), ([90, 70], 256), ([5, 200, 40], 512)])
def test_layernorm_bag_row_matches_torch(lengths, E):
    g = torch.Generator().manual_seed(7)
    rows, B = sum(lengths), len(lengths)
    x = torch.randn(rows, E, generator=g).to(DEV).requires_grad_(True)
    o = torch.randn(B, E, generator=g).to(DEV).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(E, generator=g)).to(DEV).requires_grad_(True)
    beta = (0.1 * torch.randn(E, generator=g)).to(DEV).requires_grad_(True)
    dy = torch.randn(rows, E, generator=g).to(DEV)
    segs = AttnSegs.make(lengths, [1] * B, torch.device(DEV))
    y = ops.layer_norm_bag_row(x, o, segs, gamma, beta, 1e-5, tail_rows=B)
    y.backward(dy)
    got = [t.grad.clone() for t in (x, o, gamma, beta)]
    for t in (x, o, gamma, beta):
        t.grad = None
    bag = torch.repeat_interleave(torch.arange(B, device=DEV), torch.tensor(lengths, device=DEV))
    ref = torch.nn.functional.layer_norm(x + o[bag], (E,), gamma, beta, 1e-5)
    ref.backward(dy)
    assert float((y - ref).abs().max()) <= 2e-5
    for name, a, t in zip(("dx", "do", "dgamma", "dbeta"), got, (x, o, gamma, beta)):
        assert rel_err(a, t.grad) <= 2e-5, (name, rel_err(a, t.grad))


def test_short_bags_take_the_two_launch_route():
    # [5, 200, 40]: a bag shorter than one backward workgroup's row range -> add_bag_row + layer_norm (same numbers, above)
    rpb = ops._lib.lib().mil_layernorm_bagrow_rows_per_block(245)
    assert rpb > 5
