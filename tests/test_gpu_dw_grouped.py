"""GPU: mil_linear_small_dw_grouped - the weight / bias gradients of all few-rows layers of a backward pass in ONE launch
(deferred.py) - against torch on the fusion step's 19-layer set: row counts 1 .. 64, every activation, strided operands."""
import pytest
import torch

from mil_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"
BLOCK = [(512, 512), (512, 512), (256, 512), (512, 256), (2048, 512), (512, 2048), (256, 512), (512, 256)]     # (N, K)
ALL = BLOCK * 2 + [(256, 512), (512, 256), (512, 512)]
ACTS = [0, 2, 1, 0, 2, 0, 4, 3] * 2 + [0, 0, 1]


def _dact(dy, y, act):
    if act == 0:
        return dy
    if act == 1:
        return dy * (1 - y * y)
    if act == 2:
        return dy * (y > 0)
    if act == 4:
        return dy * y * (1 - y)
    s = torch.sigmoid(1.702 * y)                       # QuickGELU: y holds the PRE-activation
    return dy * s * (1 + 1.702 * y * (1 - s))


@pytest.mark.parametrize("M", [32, 1, 7, 64])
@pytest.mark.parametrize("pad", [0, 16])
def test_grouped_weight_gradients_match_torch(M, pad):
    torch.manual_seed(M + pad)
    arr = (_lib.SmallDwDesc * len(ALL))()
    keep = []
    for d, (N, K), act in zip(arr, ALL, ACTS):
        # pad > 0: operands are column slices of wider tensors (row stride > width)
        dyb, yb, xb = (torch.randn(M, N + pad, device=DEV), torch.rand(M, N + pad, device=DEV), torch.randn(M, K + pad, device=DEV))
        dy, y, x = dyb[:, :N], yb[:, :N], xb[:, :K]
        dW, db = torch.full((N, K), 7.0, device=DEV), torch.full((N,), 7.0, device=DEV)
        keep.append((dy, y, x, dW, db, act))
        d.dy, d.yv, d.x, d.dW, d.db = dy.data_ptr(), (y.data_ptr() if act else None), x.data_ptr(), dW.data_ptr(), db.data_ptr()
        d.lddy, d.ldyv, d.ldx, d.lddw, d.act, d.M, d.N, d.K = dy.stride(0), (y.stride(0) if act else 0), x.stride(0), K, act, M, N, K
    _lib.check(_lib.lib().mil_linear_small_dw_grouped(arr, len(ALL), torch.cuda.current_stream().cuda_stream), "dw_grouped")
    torch.cuda.synchronize()
    for i, (dy, y, x, dW, db, act) in enumerate(keep):
        g = _dact(dy, y, act)
        ref = g.t() @ x
        assert float((dW - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), (i, ALL[i], act)
        assert float((db - g.sum(0)).abs().max()) <= 2e-5 * max(1.0, float(g.sum(0).abs().max())), (i, ALL[i], act)
