"""GPU: the opt-in split-bf16 products for frozen weights (csrc/linear_x.hip): error against float64 on the host next
to the fp32 MFMA GEMM's, ragged M / N, the epilogue modes, and the text tower with the option on against the default."""
from types import SimpleNamespace

import pytest
import torch

from conftest import rel_err
from mil_amd import ops, synthetic as syn

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


@pytest.mark.parametrize("pieces,bar", [(2, 2e-5), (3, 2e-6)])
def test_split_product_error_and_epilogue(pieces, bar):
    g = torch.Generator().manual_seed(pieces)
    M, N, K = 333, 200, 512
    A = torch.randn((M, K), generator=g)
    W = torch.randn((N, K), generator=g) / K ** 0.5
    b = torch.randn((N,), generator=g)
    res = torch.randn((M, N), generator=g)
    ref = (A.double() @ W.double().t() + b.double())
    Wp = ops.split_bf16(W.to(DEV), pieces)
    out = ops.gemm_split(A.to(DEV), Wp, bias=b.to(DEV), residual=res.to(DEV))
    err = float((out.cpu().double() - (ref + res.double())).abs().max() / ref.abs().max())
    f32 = ops.gemm(A.to(DEV), 0, W.to(DEV), 0, M, N, K, bias=b.to(DEV), residual=res.to(DEV))
    err32 = float((f32.cpu().double() - (ref + res.double())).abs().max() / ref.abs().max())
    assert err <= bar, (err, err32)
    # QuickGELU with the pre-activation stored, then its derivative folded into a second product's epilogue
    pre = torch.empty((M, N), device=DEV)
    h = ops.gemm_split(A.to(DEV), Wp, bias=b.to(DEV), act=3, aux=pre, aux_mode=1)
    assert float((pre.cpu().double() - ref).abs().max() / ref.abs().max()) <= bar
    hr = ref * torch.sigmoid(1.702 * ref)
    assert float((h.cpu().double() - hr).abs().max() / hr.abs().max()) <= 2 * bar
    V = torch.randn((N, K), generator=g) / K ** 0.5                      # [N, K]: dout [M, K] . V^T -> [M, N]
    d = ops.gemm_split(A.to(DEV), ops.split_bf16(V.to(DEV), pieces), aux=pre, aux_mode=2)
    sg = torch.sigmoid(1.702 * ref)
    dref = (A.double() @ V.double().t()) * (sg * (1 + 1.702 * ref * (1 - sg)))
    assert float((d.cpu().double() - dref).abs().max() / dref.abs().max()) <= 4 * bar


@pytest.mark.parametrize("pieces", [2, 3])
def test_text_tower_with_split_products_tracks_the_fp32_tower(pieces):
    from mil_amd.model.dim1.CLIP import CLIP
    torch.manual_seed(0)
    base = dict(learnablePrompt=1, n_ctx=4, clinical_features=["a", "b"], clip_layers=2)
    m32 = CLIP(SimpleNamespace(**base)).to(DEV)
    mx = CLIP(SimpleNamespace(**base, clip_gemm_pieces=pieces)).to(DEV)
    mx.load_state_dict(m32.state_dict())
    ids = syn.make_token_ids(5, 30, 3).to(DEV)                 # 90 sequences: enough rows for the tall path
    go = torch.randn((30, 3, 512), generator=torch.Generator().manual_seed(1)).to(DEV)
    outs = []
    for m in (m32, mx):
        m.zero_grad()
        f = m(ids)
        f.backward(go)
        outs.append((f.detach(), m.ctx.grad.detach().clone()))
    tol = 2e-4 if pieces == 2 else 2e-5
    assert rel_err(outs[1][0].cpu(), outs[0][0].cpu()) <= tol
    assert rel_err(outs[1][1].cpu(), outs[0][1].cpu()) <= tol
