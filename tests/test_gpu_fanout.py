"""GPU: the token-side layers sum the gradients of a multi-consumer output inside their backward kernels (ops.fan_out +
mil_linear_small_bwd_sum / mil_linear_small_ln_bwd3) and add a second input addend while staging (mil_linear_small_fwd_add) -
against plain torch autograd, which forms the same sums with elementwise adds."""
import pytest
import torch

from conftest import rel_err
from mil_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _r(g, *s, sc=1.0):
    return (sc * torch.randn(*s, generator=g)).to(DEV)


@pytest.mark.parametrize("n_cons", [2, 4, 6])
@pytest.mark.parametrize("act", ["tanh", "none"])
def test_linear_output_with_several_consumers(n_cons, act):
    g = torch.Generator().manual_seed(n_cons)
    M, K, N = 32, 512, 512
    x, W, b = _r(g, M, K), _r(g, N, K, sc=0.05), _r(g, N, sc=0.1)
    cs = [_r(g, M, N) for _ in range(n_cons)]
    lv = [t.clone().requires_grad_(True) for t in (x, W, b)]
    y = ops.linear_act(lv[0], lv[1], lv[2], act)
    hs = ops.fan_out(y, n_cons)
    assert all(h is not y for h in hs)
    sum((h * c).sum() for h, c in zip(hs, cs)).backward()
    rf = [t.clone().requires_grad_(True) for t in (x, W, b)]
    yr = torch.nn.functional.linear(rf[0], rf[1], rf[2])
    yr = torch.tanh(yr) if act == "tanh" else yr
    sum((yr * c).sum() for c in cs).backward()
    assert rel_err(y.detach(), yr.detach()) <= 2e-6
    for a, r_, nm in zip(lv, rf, ("dx", "dW", "db")):
        assert rel_err(a.grad, r_.grad) <= 2e-5, (nm, rel_err(a.grad, r_.grad))


def test_linear_with_second_input_addend():
    g = torch.Generator().manual_seed(9)
    M, K, N = 32, 512, 256
    x, x2, W, b, c, res = _r(g, M, K), _r(g, M, K), _r(g, N, K, sc=0.05), _r(g, N, sc=0.1), _r(g, M, N), _r(g, M, N)
    lv = [t.clone().requires_grad_(True) for t in (x, x2, W, b, res)]
    y = ops.linear_act(lv[0], lv[2], lv[3], "none", residual=lv[4], x2=lv[1])
    (y * c).sum().backward()
    rf = [t.clone().requires_grad_(True) for t in (x, x2, W, b, res)]
    yr = torch.nn.functional.linear(rf[0] + rf[1], rf[2], rf[3]) + rf[4]
    (yr * c).sum().backward()
    assert rel_err(y.detach(), yr.detach()) <= 2e-6
    for a, r_, nm in zip(lv, rf, ("dx", "dx2", "dW", "db", "dres")):
        assert rel_err(a.grad, r_.grad) <= 2e-5, (nm, rel_err(a.grad, r_.grad))


def test_second_addend_on_a_tall_layer_takes_the_plain_sum():
    g = torch.Generator().manual_seed(2)
    x, x2, W = _r(g, 2048, 512), _r(g, 2048, 512), _r(g, 256, 512, sc=0.05)
    y = ops.linear_act(x, W, None, "none", x2=x2)
    assert rel_err(y, torch.nn.functional.linear(x + x2, W)) <= 2e-6


@pytest.mark.parametrize("n_cons", [2, 3, 4])
def test_norm_output_with_several_consumers(n_cons):
    """lin_ln_lin's xn (the LayerNorm output later layers consume) with n consumers besides the fused C layer."""
    g = torch.Generator().manual_seed(20 + n_cons)
    M, E = 32, 512
    z, Wp, bp, resp = _r(g, M, E), _r(g, E, E, sc=0.05), _r(g, E, sc=0.1), _r(g, M, E)
    gamma, beta, Wc, bc = 1 + _r(g, E, sc=0.1), _r(g, E, sc=0.1), _r(g, 256, E, sc=0.05), _r(g, 256, sc=0.1)
    cy, cs = _r(g, M, 256), [_r(g, M, E) for _ in range(n_cons)]
    names = ("z", "Wp", "bp", "resp", "gamma", "beta", "Wc", "bc")
    lv = [t.clone().requires_grad_(True) for t in (z, Wp, bp, resp, gamma, beta, Wc, bc)]
    y, xn = ops.lin_ln_lin(lv[0], lv[1], lv[2], lv[3], lv[4], lv[5], 1e-5, None, lv[6], lv[7])
    hs = ops.fan_out(xn, n_cons)
    ((y * cy).sum() + sum((h * c).sum() for h, c in zip(hs, cs))).backward()
    rf = [t.clone().requires_grad_(True) for t in (z, Wp, bp, resp, gamma, beta, Wc, bc)]
    u = torch.nn.functional.linear(rf[0], rf[1], rf[2]) + rf[3]
    xr = torch.nn.functional.layer_norm(u, (E,), rf[4], rf[5], 1e-5)
    yr = torch.nn.functional.linear(xr, rf[6], rf[7])
    ((yr * cy).sum() + sum((xr * c).sum() for c in cs)).backward()
    for a, r_, nm in zip(lv, rf, names):
        assert rel_err(a.grad, r_.grad) <= 5e-5, (nm, rel_err(a.grad, r_.grad))


def test_fan_out_of_a_plain_tensor_is_the_tensor():
    x = torch.randn(4, 16, device=DEV, requires_grad=True)
    a, b = ops.fan_out(x, 2)
    assert a is x and b is x


@pytest.mark.parametrize("with_xn_consumer", [True, False])
def test_wide_consumer_layer_input_gradient_in_partial_sums(with_xn_consumer):
    """lin_ln_lin with a 2048-wide C layer (mlp.lin1, relu): C's input gradient comes as four partial sums over n
    (mil_linear_small_bwd_split) that the norm's backward adds while staging (mil_linear_small_ln_bwd5)."""
    g = torch.Generator().manual_seed(77)
    M, E, N = 32, 512, 2048
    z, Wp, bp, resp = _r(g, M, 256), _r(g, E, 256, sc=0.05), _r(g, E, sc=0.1), _r(g, M, E)
    gamma, beta, Wc, bc = 1 + _r(g, E, sc=0.1), _r(g, E, sc=0.1), _r(g, N, E, sc=0.05), _r(g, N, sc=0.1)
    cy, cx = _r(g, M, N), _r(g, M, E)
    names = ("z", "Wp", "bp", "resp", "gamma", "beta", "Wc", "bc")
    lv = [t.clone().requires_grad_(True) for t in (z, Wp, bp, resp, gamma, beta, Wc, bc)]
    y, xn = ops.lin_ln_lin(lv[0], lv[1], lv[2], lv[3], lv[4], lv[5], 1e-5, None, lv[6], lv[7], "relu")
    ((y * cy).sum() + ((xn * cx).sum() if with_xn_consumer else 0.0)).backward()
    rf = [t.clone().requires_grad_(True) for t in (z, Wp, bp, resp, gamma, beta, Wc, bc)]
    u = torch.nn.functional.linear(rf[0], rf[1], rf[2]) + rf[3]
    xr = torch.nn.functional.layer_norm(u, (E,), rf[4], rf[5], 1e-5)
    yr = torch.relu(torch.nn.functional.linear(xr, rf[6], rf[7]))
    ((yr * cy).sum() + ((xr * cx).sum() if with_xn_consumer else 0.0)).backward()
    assert rel_err(y.detach(), yr.detach()) <= 2e-6
    for a, r_, nm in zip(lv, rf, names):
        assert rel_err(a.grad, r_.grad) <= 5e-5, (nm, rel_err(a.grad, r_.grad))
