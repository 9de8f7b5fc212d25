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
            o.step()           # `extra` never receives a gradient: torch.optim.Adam skips it, and so does FlatAdam
    for (k, a), b in zip(ref.named_parameters(), ours.parameters()):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max())), k
    assert all(p.data_ptr() >= o_our.flat.data_ptr() for p in ours.parameters())     # still views of the flat buffer
    sd = o_our.state_dict()
    assert sd["step"] == 6
    torch.manual_seed(3)
    assert torch.equal(ours.extra.detach().cpu(), _model().extra.detach().cpu())      # untouched: no weight-decay drift


def test_zero_gradient_mark_keeps_upstream_weight_decay():
    """A parameter a fast path leaves out of the graph although upstream it gets a dense ZERO gradient (q / k projections
    of a one-key attention) is marked `_mil_zero_grad`: torch.optim.Adam decays it, so FlatAdam must too."""
    ref = _model()
    ours = copy.deepcopy(ref)
    ours.extra._mil_zero_grad = True
    o_ref = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=1e-2)
    o_our = FlatAdam(ours.parameters(), lr=1e-3, weight_decay=1e-2)
    x = torch.randn((11, 37), device=DEV)
    for _ in range(3):
        for m, o in ((ref, o_ref), (ours, o_our)):
            o.zero_grad()
            m(x).square().mean().backward()
            if m is ref:
                m.extra.grad = torch.zeros_like(m.extra)
            o.step()
    assert float((ref.extra - ours.extra).abs().max()) <= 2e-6
    assert float((ours.extra - _model().extra).abs().max()) > 1e-4                     # it did move


def test_grad_slot_is_handed_out_once_per_backward_pass():
    """A weight used by TWO nodes in one step: only the first backward may write the flat slot in place, the second
    gets a fresh tensor and autograd accumulates, so p.grad = g1 + g2 (not 2 g2)."""
    from mil_amd import ops
    torch.manual_seed(2)
    lin = torch.nn.Linear(512, 256).to(DEV)
    ref = copy.deepcopy(lin)
    opt = FlatAdam(list(lin.parameters()), lr=1e-3)
    x1, x2 = torch.randn((32, 512), device=DEV), torch.randn((32, 512), device=DEV)
    opt.zero_grad()
    (ops.linear_act(x1, lin.weight, lin.bias, "tanh").sum() + 3.0 * ops.linear_act(x2, lin.weight, lin.bias, "tanh").sum()).backward()
    (torch.tanh(ref(x1)).sum() + 3.0 * torch.tanh(ref(x2)).sum()).backward()
    assert float((lin.weight.grad - ref.weight.grad).abs().max()) <= 1e-3 * float(ref.weight.grad.abs().max())
    assert float((lin.bias.grad - ref.bias.grad).abs().max()) <= 1e-3 * float(ref.bias.grad.abs().max())


def test_linear_gradients_land_in_the_flat_slots_without_a_copy():
    """ops.linear_act's backward writes dW / db into FlatAdam's slots and autograd adopts those tensors as .grad."""
    from mil_amd import ops
    torch.manual_seed(1)
    lin = torch.nn.Linear(512, 256).to(DEV)
    big = torch.nn.Linear(512, 128).to(DEV)
    opt = FlatAdam(list(lin.parameters()) + list(big.parameters()), lr=1e-3)
    xs = torch.randn((32, 512), device=DEV)           # few-rows path
    xb = torch.randn((300, 512), device=DEV)          # tiled path
    opt.zero_grad()
    (ops.linear_act(xs, lin.weight, lin.bias, "tanh").sum() + ops.linear_act(xb, big.weight, big.bias, "relu").sum()).backward()
    for p in list(lin.parameters()) + list(big.parameters()):
        assert p.grad is not None and p.grad.data_ptr() == p._mil_grad.data_ptr()
    ref = torch.tanh(xs @ lin.weight.t() + lin.bias)
    assert float((lin.bias.grad - (1 - ref * ref).sum(0)).abs().max()) <= 1e-4


def test_flat_sgd_tracks_torch_sgd():
    """optim.FlatSGD (mil_sgd_step) against torch.optim.SGD(lr=1e-3, weight_decay=1e-7), the learnable-prompt optimizer
    (train_ddp.py:103-108)."""
    from mil_amd.optim import FlatSGD
    ref = _model()
    ours = copy.deepcopy(ref)
    o_ref = torch.optim.SGD(ref.parameters(), lr=1e-2, weight_decay=1e-3)
    o_our = FlatSGD(ours.parameters(), lr=1e-2, weight_decay=1e-3)
    g = torch.Generator().manual_seed(0)
    for step in range(5):
        x = torch.randn((11, 37), generator=g).to(DEV)
        for m, o in ((ref, o_ref), (ours, o_our)):
            o.zero_grad()
            m(x).square().mean().backward()
            o.step()           # `extra` has no gradient: skipped by torch.optim.SGD and by FlatSGD alike
    for (k, a), b in zip(ref.named_parameters(), ours.parameters()):
        assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(a.abs().max())), k
    assert o_our.exp_avg.numel() == 0


def test_flat_adam_large_buffer_path():
    """More than 2^21 parameters take the four-float4-per-thread Adam kernel (and its scalar tail)."""
    torch.manual_seed(4)
    ref = torch.nn.Linear(2048, 1101).to(DEV)            # 2 255 949 parameters: not a multiple of the 4096-element block
    ours = copy.deepcopy(ref)
    o_ref = torch.optim.Adam(ref.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-7)
    o_our = FlatAdam(ours.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-7)
    g = torch.Generator().manual_seed(0)
    for step in range(3):
        x = torch.randn((7, 2048), generator=g).to(DEV)
        for m, o in ((ref, o_ref), (ours, o_our)):
            o.zero_grad()
            m(x).square().mean().backward()
            o.step()
    for (k, a), b in zip(ref.named_parameters(), ours.parameters()):
        assert float((a.detach() - b.detach()).abs().max()) <= 2e-6 * max(1.0, float(a.detach().abs().max())), k


def test_deferred_grouped_weight_gradients_equal_the_per_layer_ones(monkeypatch):
    """deferred.py: with FlatAdam's gradient slots the few-rows layers launch dx alone and one grouped launch forms every
    dW / db at the end of the backward pass - same numbers as the per-layer launches (MIL_DEFER_DW=0), bit for bit."""
    import torch.nn as nn
    from mil_amd import ops
    from mil_amd.optim import FlatAdam

    def run(defer):
        monkeypatch.setenv("MIL_DEFER_DW", "1" if defer else "0")
        torch.manual_seed(0)
        layers = [nn.Linear(512, 512), nn.Linear(512, 2048), nn.Linear(2048, 512), nn.Linear(512, 256)]
        layers = [l.to("cuda") for l in layers]
        acts = ["tanh", "relu", "none", "none"]
        opt = FlatAdam([p for l in layers for p in l.parameters()], lr=1e-3)
        x = torch.randn(32, 512, generator=torch.Generator().manual_seed(1)).to("cuda").requires_grad_(True)
        h = x
        for l, a in zip(layers, acts):
            h = ops.linear_act(h, l.weight, l.bias, a)
        opt.zero_grad()
        (h * h).sum().backward()
        torch.cuda.synchronize()
        grads = [p.grad.clone() for l in layers for p in l.parameters()] + [x.grad.clone()]
        in_slot = all(p.grad.data_ptr() == p._mil_grad.data_ptr() for l in layers for p in l.parameters())
        return grads, in_slot

    g1, slot1 = run(True)
    g0, slot0 = run(False)
    assert slot1 and slot0
    for a, b in zip(g1, g0):
        assert torch.equal(a, b)


def test_one_launch_multi_range_adam_equals_the_per_range_launches():
    """mil_adam_step_dev_segs (round 4: every live range of the flat buffer and the step-counter advance in one launch) against
    one mil_adam_step_dev launch per range: bit-identical parameters and moments over several steps, counter advanced once
    per step, sign-off word back at zero (what a hipGraph replay needs)."""
    from mil_amd import ops
    dev = torch.device("cuda")
    n = 3 * 4096 * 40 + 1236
    g = torch.Generator().manual_seed(5)
    base = [torch.randn(n, generator=g).to(dev) for _ in range(2)]
    segs = [(0, 5000), (8192, 8192 + 4096 * 7), (40000, 40000), (50000, n)]           # short tail, whole blocks, empty, long
    lr_dev = torch.full((1,), 1e-3, device=dev)

    def run(one_launch):
        p, m, v = base[0].clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        ctr, done = torch.zeros(1, device=dev, dtype=torch.int32), torch.zeros(1, device=dev, dtype=torch.int32)
        for step in range(4):
            grad = base[1] * (1.0 + 0.1 * step)
            if one_launch:
                ops.adam_step_dev_segs(p, grad, m, v, segs, ctr, lr_dev, done, grad_scale=0.5)
            else:
                live = [(a, b) for a, b in segs if b > a]
                for i, (a, b) in enumerate(live):
                    ops.adam_step_dev(p[a:b], grad[a:b], m[a:b], v[a:b], ctr, lr_dev, grad_scale=0.5, inc=(i == len(live) - 1))
        torch.cuda.synchronize()
        return p, m, v, int(ctr.item()), int(done.item())

    pa, ma, va, ca, da = run(True)
    pb, mb, vb, cb, _ = run(False)
    assert ca == cb == 4 and da == 0
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert torch.equal(pa[5000:8192], base[0][5000:8192])                             # outside every range: untouched
