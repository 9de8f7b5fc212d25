"""GPU: CLIP-as-loss (reference utils.py:247-284, the aggregator_clip.py training path).  utils.py itself cannot be
imported offline (it needs `clip.load`), so the pin is its arithmetic restated with the same torch ops
(oracle.clip_contrastive_loss: matmul + CrossEntropyLoss with probability targets): "pinned by torch"."""
import pytest
import torch

from conftest import rel_err
from mil_amd import ops
from oracle import mil_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("b,F,E", [(1, 9, 512), (3, 9, 512), (32, 9, 512), (8, 2, 64)])
def test_clip_contrastive_loss_fwd_bwd(b, F, E):
    g = torch.Generator().manual_seed(b * 10 + F)
    out = torch.randn(b, E, generator=g) * 0.3
    feat = torch.randn(b, F, E, generator=g) * 0.3
    od = out.cuda().requires_grad_(True)
    loss = ops.clip_contrastive_loss(od, feat.cuda())
    (loss * 1.7).backward()
    orf = out.clone().requires_grad_(True)
    ref = orc.clip_contrastive_loss(orf, feat)
    (ref * 1.7).backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert rel_err(od.grad.cpu(), orf.grad) <= 1e-5 or float(orf.grad.abs().max()) < 1e-7
