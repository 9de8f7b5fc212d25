"""GPU: the optional 'textCosSim' loss term - torch.nn.CosineEmbeddingLoss between the two text-aligned tokens x_CT2CI and
x_Pth2CI (reference train_ddp.py:102,266,325-329) - as one fused forward + backward launch, against vectors produced by
torch's own op on the fused_ct_pth tokens (tests/golden/cossim_ct_pth.npz), and the train-loop return contract
`([out, out, out], [CT2CI, Pth2CI], None)` of train_ddp.py:300."""
from types import SimpleNamespace

import pytest
import torch

from conftest import load_golden, rel_err
from mil_amd import ops, synthetic as syn
from mil_amd.model.utils import get_model

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("case", ["tok", "rnd"])
def test_cosine_embedding_loss_vs_torch_vectors(case):
    g = load_golden("cossim_ct_pth")
    a = g[case + ".x1"].to(DEV).requires_grad_(True)
    b = g[case + ".x2"].to(DEV).requires_grad_(True)
    loss = ops.cosine_embedding_loss(a, b)
    (loss * 1.0).backward()
    assert abs(float(loss.detach()) - float(g[case + ".loss"])) <= 1e-6
    assert rel_err(a.grad.cpu(), g[case + ".dx1"]) <= 1e-5 and rel_err(b.grad.cpu(), g[case + ".dx2"]) <= 1e-5


def test_train_contract_and_cossim_term_on_the_ct_pathology_branch():
    """train_contract=1: the 3-tuple the reference's training loop unpacks (train_ddp.py:300,319-329); the BCE on outputs[0]
    plus the CosSim term on the two tokens back-propagates through the whole module and matches the oracle's value."""
    from oracle import mil_oracle as orc
    g = load_golden("fused_ct_pth")
    seed = int(g["seed"])
    B, N, P, D, hw, clayers = [int(v) for v in g["cfg"]]
    args = SimpleNamespace(modality=["CT", "pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL",
                           num_classes=2, learnablePrompt=0, alignment_base="CI", model_CT="resnetMC3_18", clip_layers=clayers,
                           train_contract=1)
    model = get_model(args)
    model.load_state_dict(syn.fused_params(seed, "TwoWayTransformer_Both", clip_layers=clayers, with_ct=True), strict=False)
    model = model.to(DEV).eval()
    x = syn.make_bags(seed + 3, B, N, 768).to(DEV)
    ids = syn.make_token_ids(seed + 4, B, P).to(DEV)
    y = syn.make_labels(seed + 5, B).to(DEV)
    ct = syn.make_ct_map(seed + 6, B, D, hw).to(DEV)
    outs, toks, attns = model([ct, x], ids)
    assert isinstance(outs, list) and len(outs) == 3 and isinstance(toks, list) and len(toks) == 2 and attns is None
    assert all(o is outs[0] for o in outs) and toks[0].shape == (B, P, 512)
    assert float((model.last_logits.detach().cpu() - g["logits"]).abs().max()) <= 2e-5
    cs = ops.cosine_embedding_loss(toks[0].squeeze(1), toks[1].squeeze(1))
    want = orc.cosine_embedding_loss(g["x_CT2CI"].squeeze(1), g["x_Pth2CI"].squeeze(1))
    assert abs(float(cs) - float(want)) <= 1e-5
    loss = torch.nn.BCELoss()(outs[0], y) + cs                       # train_ddp.py:323-329
    loss.backward()
    gq = model.fc_CI2CT[0].weight.grad
    assert gq is not None and bool(torch.isfinite(gq).all()) and float(gq.abs().max()) > 0
