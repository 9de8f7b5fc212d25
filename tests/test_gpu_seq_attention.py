"""GPU: whole-sequence self-attention ("seq" form, the CLIP text blocks: clip/model.py:171-184 with the causal mask of
:324-330) against torch fp32 on the host - forward, lse and the backward, ragged sequence lengths up to 96 tokens,
both head widths, causal and full."""
import pytest
import torch

from conftest import rel_err
from mil_amd import ops
from mil_amd.segments import AttnSegs

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _ref(q, k, v, lens, H, causal):
    outs, off = [], 0
    C = q.shape[1] // H
    for n in lens:
        qs, ks, vs = (t[off:off + n].reshape(n, H, C).transpose(0, 1) for t in (q, k, v))       # [H, n, C]
        s = qs @ ks.transpose(1, 2) / C ** 0.5
        if causal:
            s = s + torch.full((n, n), float("-inf")).triu(1)
        outs.append((torch.softmax(s, -1) @ vs).transpose(0, 1).reshape(n, H * C))
        off += n
    return torch.cat(outs, 0)


@pytest.mark.parametrize("C", [32, 64])
@pytest.mark.parametrize("causal", [True, False])
def test_seq_attention_matches_torch(C, causal):
    H, lens = 8, [77, 20, 33, 96, 64, 17]
    I, R = H * C, sum(lens)
    g = torch.Generator().manual_seed(C + int(causal))
    q, k, v, go = (torch.randn((R, I), generator=g) for _ in range(4))
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    out_ref = _ref(qr, kr, vr, lens, H, causal)
    out_ref.backward(go)
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    segs = AttnSegs.make(lens, lens, DEV)
    out = ops.attention_rows(qd, kd, vd, segs, H, causal=causal)
    out.backward(go.to(DEV))
    assert rel_err(out.detach().cpu(), out_ref.detach()) <= 2e-6
    assert rel_err(qd.grad.cpu(), qr.grad) <= 1e-5
    assert rel_err(kd.grad.cpu(), kr.grad) <= 1e-5
    assert rel_err(vd.grad.cpu(), vr.grad) <= 1e-5
