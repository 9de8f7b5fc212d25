"""GPU: the token-side (few rows) nn.Linear entry points against torch fp32 on the host, forward and the one-launch
backward, over every activation, row counts on both sides of the 32-row tile and ragged N / K."""
import pytest
import torch

from conftest import rel_err
from mil_amd import ops

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")
ACTS = {"none": lambda v: v, "tanh": torch.tanh, "relu": torch.relu, "quickgelu": lambda v: v * torch.sigmoid(1.702 * v)}


@pytest.mark.parametrize("M,N,K", [(1, 512, 512), (7, 256, 512), (32, 2048, 512), (32, 512, 2048), (33, 512, 256),
                                   (64, 48, 32), (20, 1536, 512)])
@pytest.mark.parametrize("act", ["none", "tanh", "relu", "quickgelu"])
def test_small_linear_matches_torch(M, N, K, act):
    g = torch.Generator().manual_seed(M * 131 + N * 7 + K)
    x = torch.randn((M, K), generator=g)
    W = torch.randn((N, K), generator=g) / K ** 0.5
    b = torch.randn((N,), generator=g) * 0.1
    res = torch.randn((M, N), generator=g) if act == "none" else None
    go = torch.randn((M, N), generator=g)
    xr, Wr, br = (t.clone().requires_grad_(True) for t in (x, W, b))
    yr = ACTS[act](xr @ Wr.t() + br)
    if res is not None:
        yr = yr + res
    yr.backward(go)
    xd, Wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, W, b))
    assert ops._small_ok(M, N, K, xd, Wd)
    yd = ops.linear_act(xd, Wd, bd, act, residual=res.to(DEV) if res is not None else None)
    yd.backward(go.to(DEV))
    assert rel_err(yd.detach().cpu(), yr.detach()) <= 2e-6
    assert rel_err(xd.grad.cpu(), xr.grad) <= 5e-6
    assert rel_err(Wd.grad.cpu(), Wr.grad) <= 5e-6
    assert rel_err(bd.grad.cpu(), br.grad) <= 5e-6


def test_small_linear_partial_grads_and_limits():
    g = torch.Generator().manual_seed(5)
    x = torch.randn((32, 512), generator=g).to(DEV)
    W = (torch.randn((256, 512), generator=g) / 22).to(DEV).requires_grad_(True)
    y = ops.linear_act(x, W, None, "relu")                  # no bias, input without gradient
    y.sum().backward()
    ref = (torch.relu(x.cpu() @ W.detach().cpu().t()) > 0).float().t() @ x.cpu()
    assert rel_err(W.grad.cpu(), ref) <= 5e-6
    from mil_amd import _lib
    rc = _lib.lib().mil_linear_small_fwd(x.data_ptr(), 512, W.data_ptr(), 512, None, 0, None, 0, y.data_ptr(), 256,
                                         65, 256, 512, None)
    assert rc == -22                                        # more rows than MIL_SMALL_ROWS: refused, not truncated


def test_cpp_extension_binding_equals_the_ctypes_binding():
    """The same C entry through csrc/torch_shim.cpp (torch cpp_extension) and through ctypes: identical bits."""
    import os
    from mil_amd import _lib
    os.environ["MIL_TORCH_SHIM"] = "1"
    _lib._shim = False
    try:
        sh = _lib.shim()
    finally:
        os.environ.pop("MIL_TORCH_SHIM", None)
    if sh is None:
        _lib._shim = False
        pytest.skip("torch shim not built")
    gen = torch.Generator().manual_seed(3)
    x, W, b = (torch.randn(s, generator=gen).cuda() for s in ((32, 512), (256, 512), (256,)))
    dy = torch.randn((32, 256), generator=gen).cuda()
    saved = _lib._shim
    try:
        y1 = ops.linear_small_fwd(x, W, b, 1)
        g1 = ops.linear_small_bwd(dy, y1, 1, x, W, True, True, True)
        _lib._shim = None
        y0 = ops.linear_small_fwd(x, W, b, 1)
        g0 = ops.linear_small_bwd(dy, y0, 1, x, W, True, True, True)
    finally:
        _lib._shim = False
    assert torch.equal(y0, y1) and all(torch.equal(a, c) for a, c in zip(g0, g1))
