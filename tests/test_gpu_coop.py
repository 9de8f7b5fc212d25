"""GPU: learnable-context (CoOp) branch of the clinical-text extractor (reference model/dim1/CLIP.py:29-62): forward and
the backward THROUGH the frozen text tower (GEMM dx, LayerNorm, causal attention, QuickGELU kernels), against golden
vectors from the reference's own CLIP class."""
from types import SimpleNamespace

import pytest
import torch

from conftest import load_golden, rel_err
from mil_amd import synthetic as syn
from mil_amd.model.dim1 import CLIP
from mil_amd.model.utils import get_model

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("tag", ["coop_small", "coop_w512"])
def test_learnable_prompts_vs_golden(tag):
    g = load_golden(tag)
    width, layers, vocab, heads, embed, P, n_ctx = [int(v) for v in g["cfg"]]
    seed = int(g["seed"])
    p = syn.clip_text_params(seed, width=width, layers=layers, vocab=vocab, embed=embed)
    args = SimpleNamespace(learnablePrompt=1, n_ctx=n_ctx, clinical_features=["f"] * (P - 1), clip_vocab=vocab,
                           clip_width=width, clip_heads=heads, clip_layers=layers, clip_embed=embed)
    m = CLIP(args)
    m.model.load_state_dict({k[len("clinic_extractor.model."):]: v for k, v in p.items()}, strict=False)
    gen = torch.Generator().manual_seed(seed + 2)
    ctx0 = torch.randn((P, n_ctx, width), generator=gen) * 0.02
    with torch.no_grad():
        m.ctx.copy_(ctx0)
    m = m.to(DEV)
    ids = syn.make_token_ids(seed + 1, 1, P, vocab=vocab).to(DEV)              # [1, P, 77]
    out = m(ids)
    go = torch.randn((P, embed), generator=gen).to(DEV)
    (out[0] * go).sum().backward()
    assert rel_err(out[0].detach().cpu(), g["out"]) <= 5e-5
    assert rel_err(m.ctx.grad.cpu(), g["dctx"]) <= 5e-4
    assert all(q.grad is None for q in m.model.parameters())                   # the tower stays frozen


def test_aggregator_with_learnable_prompts_trains_ctx():
    args = SimpleNamespace(modality=["pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL",
                           num_classes=2, learnablePrompt=1, n_ctx=4, clinical_features=["a", "b"], clip_layers=1)
    model = get_model(args).to(DEV).eval()
    x = syn.make_bags(3, 2, 64, 768).to(DEV)
    ids = syn.make_token_ids(4, 2, 3).to(DEV)
    y = syn.make_labels(5, 2).to(DEV)
    prob, q = model([x], ids)
    torch.nn.BCELoss()(prob, y).backward()
    gctx = model.clinic_extractor.ctx.grad
    assert gctx is not None and float(gctx.abs().max()) > 0 and torch.isfinite(gctx).all()
    assert tuple(q.shape) == (2, 3, 512)
