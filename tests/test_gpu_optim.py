"""GPU: optim.FlatAdam (one flat buffer, one launch) against torch.optim.Adam with the reference's settings
(train_ddp.py:115-118), including a parameter that never receives a gradient and the device-counted variant."""
import copy

import pytest
import torch

from mil_amd.optim import FlatAdam

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _model():
    torch.manual_seed(3)
    m = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Tanh(), torch.nn.Linear(19, 5)).to(DEV)
    m.extra = torch.nn.Parameter(torch.randn(7, device=DEV))          # numel not a multiple of 4, never used
    return m


@pytest.mark.parametrize("counted", [False, True])
def test_flat_adam_tracks_torch_adam(counted):
    ref, ours = _model(), None
    ours = copy.deepcopy(ref)
    lr = 1e-3
    o_ref = torch.optim.Adam(ref.parameters(), lr=lr, betas=(0.9, 0.999), weight_decay=1e-7)
    o_our = FlatAdam(ours.parameters(), lr=lr, betas=(0.9, 0.999), weight_decay=1e-7, counted=counted)
    g = torch.Generator().manual_seed(0)
    for step in range(6):
        x = torch.randn((11, 37), generator=g).to(DEV)
        for m, o in ((ref, o_ref), (ours, o_our)):
            o.zero_grad()
            m(x).square().mean().backward()
            if m is ref and m.extra.grad is None:
                m.extra.grad = torch.zeros_like(m.extra)       # upstream semantics of a dense zero gradient
            o.step()
    for (k, a), b in zip(ref.named_parameters(), ours.parameters()):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max())), k
    assert all(p.data_ptr() >= o_our.flat.data_ptr() for p in ours.parameters())     # still views of the flat buffer
    sd = o_our.state_dict()
    assert sd["step"] == 6
