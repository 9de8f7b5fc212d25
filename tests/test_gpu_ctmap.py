"""GPU: the CT-token side of the multi-modal bag.  The CT encoders are out of scope, their output - the feature map
[B, 512, 160, h, w] of model/aggregator.py:139-140 - is accepted precomputed: map -> tokens (sam/transformer.py:86-98),
TwoWayTransformer on it, and the 4-segment bag of aggregator.py:155-173, against fixtures made by the reference's own
TwoWayTransformer / ABMIL on random maps (oracle/gen_golden.py ct)."""
from types import SimpleNamespace

import pytest
import torch

from conftest import check_grad, load_golden, rel_err
from mil_amd import ops, synthetic as syn
from mil_amd.model.sam.transformer import TwoWayTransformer
from mil_amd.model.utils import get_model
from oracle import mil_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("model_CT,shape", [("resnetMC3_18", (2, 512, 160, 3, 5)), ("medicalNet", (1, 512, 6, 4, 4)),
                                            ("resnetMC3_18", (1, 512, 160, 14, 14))])
def test_ct_map_tokens(model_CT, shape):
    ct = torch.randn(shape, generator=torch.Generator().manual_seed(1))
    rows, T = ops.ct_map_tokens(ct.to(DEV), model_CT)
    want = orc.ct_map_tokens(ct, model_CT)
    assert T == want.shape[1]
    assert float((rows.view(shape[0], T, 512).cpu() - want).abs().max()) <= 2e-6


@pytest.mark.parametrize("tag", ["twoway_ctmap_T1", "twoway_ctmap_T10"])
def test_twoway_transformer_on_a_ct_map_vs_reference(tag):
    g = load_golden(tag)
    seed = int(g["seed"])
    T, D, hw = [int(v) for v in g["shape"]]
    name = "TwoWayTransformer_CT"
    p = syn.twoway_params(seed, name)
    m = TwoWayTransformer(args=SimpleNamespace(alignment_base="CI", model_CT="resnetMC3_18"), depth=2, embedding_dim=512,
                          num_heads=8, mlp_dim=2048)
    m.load_state_dict({k[len(name) + 1:]: v for k, v in p.items()})
    m = m.to(DEV).eval()
    ct = syn.make_ct_map(seed + 1, 1, D, hw).to(DEV)
    gen = torch.Generator().manual_seed(seed + 2)
    pt = torch.randn((1, T, 512), generator=gen).to(DEV).requires_grad_(True)
    q, k = m(ct, orc.sinusoidal_pe(D, 512).unsqueeze(0).to(DEV), pt)
    gq = torch.randn((1, T, 512), generator=gen).to(DEV)
    gk = torch.randn((1, D, 512), generator=gen).to(DEV)
    ((q * gq).sum() + (k * gk).sum()).backward()
    assert rel_err(q[0].detach().cpu(), g["queries"]) <= 5e-5 and rel_err(k[0].detach().cpu(), g["keys"]) <= 5e-5
    assert rel_err(pt.grad[0].cpu(), g["dpoint"]) <= 5e-4
    for n, prm in m.named_parameters():
        gn = float(g["g." + name + "." + n + ".norm"])
        got = prm.grad if prm.grad is not None else torch.zeros_like(prm)
        if gn == 0.0:
            assert float(got.abs().max()) <= 1e-10, n
        else:
            check_grad("g." + name + "." + n, got, g, 1e-3)


@pytest.mark.parametrize("tag", ["twoway_ctbase_N64", "twoway_ctbase_D160"])
def test_twoway_transformer_alignment_base_ct_vs_reference(tag):
    """`--alignment_base CT` (sam/transformer.py:78-86): the 5-D CT map is the POINT embedding - its D tokens are the queries of
    every attention (more than 16 per bag: the general rows kernels, mil_attn_rows_bwd_general; D = 160 also exceeds the
    whole-sequence self-attention kernel) - against the reference's own TwoWayTransformer (oracle/gen_golden.py ctbase)."""
    g = load_golden(tag)
    seed = int(g["seed"])
    N, D, hw = [int(v) for v in g["shape"]]
    name = "TwoWayTransformer_Pth"
    p = syn.twoway_params(seed, name)
    m = TwoWayTransformer(args=SimpleNamespace(alignment_base="CT", model_CT="resnetMC3_18"), depth=2, embedding_dim=512,
                          num_heads=8, mlp_dim=2048)
    m.load_state_dict({k[len(name) + 1:]: v for k, v in p.items()})
    m = m.to(DEV).eval()
    ct = syn.make_ct_map(seed + 1, 1, D, hw).to(DEV)
    gen = torch.Generator().manual_seed(seed + 2)
    img = torch.randn((1, N, 512), generator=gen).to(DEV).requires_grad_(True)
    q, k = m(img, orc.sinusoidal_pe(N, 512).unsqueeze(0).to(DEV), ct)
    assert q.shape == (1, D, 512) and k.shape == (1, N, 512)
    gq = torch.randn((1, D, 512), generator=gen).to(DEV)
    gk = torch.randn((1, N, 512), generator=gen).to(DEV)
    ((q * gq).sum() + (k * gk).sum()).backward()
    assert rel_err(q[0].detach().cpu(), g["queries"]) <= 5e-5 and rel_err(k[0].detach().cpu(), g["keys"]) <= 5e-5
    assert rel_err(img.grad[0].cpu(), g["dimage"]) <= 5e-4
    for n, prm in m.named_parameters():
        gn = float(g["g." + name + "." + n + ".norm"])
        got = prm.grad if prm.grad is not None else torch.zeros_like(prm)
        if gn == 0.0:
            assert float(got.abs().max()) <= 1e-10, n
        else:
            check_grad("g." + name + "." + n, got, g, 1e-3)


def test_alignment_base_ct_at_a_realistic_bag_length_vs_the_oracle():
    """the general rows kernels (a thread per (row, head) looping over the other side) were pinned at N <= 200 only.
    4096 patches x 160 CT-token queries: long softmax sums (expf(s - lse) over 4096 keys), 160 x 4096 dependent iterations
    in the key / value backward - forward and the image gradient against the oracle's TwoWayTransformer restatement, inside a
    bounded run time (the kernels are correct-not-fast: no shipped run takes --alignment_base CT)."""
    import time
    seed, N, D, hw = 91, 4096, 160, 2
    name = "TwoWayTransformer_Pth"
    p = syn.twoway_params(seed, name)
    m = TwoWayTransformer(args=SimpleNamespace(alignment_base="CT", model_CT="resnetMC3_18"), depth=2, embedding_dim=512,
                          num_heads=8, mlp_dim=2048)
    m.load_state_dict({k[len(name) + 1:]: v for k, v in p.items()})
    m = m.to(DEV).eval()
    ct = syn.make_ct_map(seed + 1, 1, D, hw)
    gen = torch.Generator().manual_seed(seed + 2)
    img_c = torch.randn((N, 512), generator=gen)
    gq = torch.randn((D, 512), generator=gen)
    gk = torch.randn((N, 512), generator=gen)
    img = img_c.unsqueeze(0).to(DEV).requires_grad_(True)
    torch.cuda.synchronize()
    t0 = time.time()
    q, k = m(img, orc.sinusoidal_pe(N, 512).unsqueeze(0).to(DEV), ct.to(DEV))
    ((q[0] * gq.to(DEV)).sum() + (k[0] * gk.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    took = time.time() - t0
    leaf = img_c.clone().requires_grad_(True)
    qo, ko = orc.twoway_transformer(leaf, orc.sinusoidal_pe(N, 512), orc.ct_map_tokens(ct)[0], p, name)
    ((qo * gq).sum() + (ko * gk).sum()).backward()
    assert rel_err(q[0].detach().cpu(), qo.detach()) <= 5e-5 and rel_err(k[0].detach().cpu(), ko.detach()) <= 5e-5
    assert rel_err(img.grad[0].cpu(), leaf.grad) <= 5e-4
    assert took < 20.0, took                              # measured: a fraction of a second; guards against a spill / timeout regression


def _args(**kw):
    a = dict(modality=["CT", "pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL", num_classes=2,
             learnablePrompt=0, alignment_base="CI", model_CT="resnetMC3_18")
    a.update(kw)
    return SimpleNamespace(**a)


@pytest.mark.parametrize("flat", [False, True])
def test_ct_plus_pathology_module_vs_reference_wiring(flat):
    """aggregator(args) with modality ['CT', 'pathology']: returns (prob, x_CT2CI, x_Pth2CI) as aggregator.py:202-203.
    flat=True: under optim.FlatAdam the few-rows layers defer their weight gradients to one grouped launch; the layers of
    TwoWayTransformer_Both run TWICE per forward (aggregator.py:160-168), so both uses' gradients must meet in the flat slot
    (the first use's gradient was lost)."""
    g = load_golden("fused_ct_pth")
    seed = int(g["seed"])
    B, N, P, D, hw, clayers = [int(v) for v in g["cfg"]]
    p = syn.fused_params(seed, "TwoWayTransformer_Both", clip_layers=clayers, with_ct=True)
    model = get_model(_args(clip_layers=clayers))
    missing, unexpected = model.load_state_dict(p, strict=False)
    assert not unexpected, unexpected
    model = model.to(DEV).eval()
    opt = None
    if flat:
        from mil_amd.optim import FlatAdam
        opt = FlatAdam([q for q in model.parameters() if q.requires_grad])
        opt.grad.fill_(123.0)                       # stale slot contents must not leak into any gradient
        opt.zero_grad()
    x = syn.make_bags(seed + 3, B, N, 768).to(DEV)
    ids = syn.make_token_ids(seed + 4, B, P).to(DEV)
    y = syn.make_labels(seed + 5, B).to(DEV)
    ct = syn.make_ct_map(seed + 6, B, D, hw).to(DEV)
    prob, q_ct, q_p = model([ct, x], ids)
    loss = torch.nn.BCELoss()(prob, y)
    loss.backward()
    if flat:
        opt.gather()
    assert float((model.last_logits.detach().cpu() - g["logits"]).abs().max()) <= 2e-5         # bar: 1e-3
    assert torch.equal(prob.detach().cpu().argmax(-1), g["prob"].argmax(-1))
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5
    assert rel_err(q_ct.detach().cpu(), g["x_CT2CI"]) <= 5e-5 and rel_err(q_p.detach().cpu(), g["x_Pth2CI"]) <= 5e-5
    params = dict(model.named_parameters())
    for k in p:
        if k.startswith("clinic_extractor."):
            continue
        gn = float(g["g." + k + ".norm"])
        got = params[k].grad if params[k].grad is not None else torch.zeros_like(params[k])
        if flat and params[k].grad is not None:
            got = params[k]._mil_grad               # what Adam will read: the flat buffer's slot after gather()
        if gn == 0.0:
            assert float(got.abs().max()) <= 1e-10, k
        else:
            check_grad("g." + k, got, g, 1e-3)


def test_ct_only_module_vs_oracle():
    """modality ['CT']: TwoWayTransformer_CT, bag = [x_CT2CI | x_CI2CT] (aggregator.py:176-184), returns (prob, x_CT2CI)."""
    torch.manual_seed(5)
    model = get_model(_args(modality=["CT"], clip_layers=1)).to(DEV).eval()
    B, D, hw = 2, 160, 2
    ct = syn.make_ct_map(9, B, D, hw)
    ids = syn.make_token_ids(10, B, 1)
    prob, q = model([ct.to(DEV)], ids.to(DEV))
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    for b in range(B):
        with torch.no_grad():
            t = orc.clip_encode_text(ids[b], sd, 8)
            tok = orc.ct_map_tokens(ct[b:b + 1])[0]
            a, c_ = orc.twoway_transformer(tok, orc.sinusoidal_pe(D, 512), orc.linear_tanh(t, sd["fc_CI2CT.0.weight"], sd["fc_CI2CT.0.bias"]),
                                           sd, "TwoWayTransformer_CT")
            M, _, _ = orc.abmil_forward(torch.cat([a, c_], 0), sd)
            z, pr = orc.head_forward(M, sd)
        assert float((model.last_logits[b].detach().cpu() - z[0]).abs().max()) <= 2e-5
        assert rel_err(q[b].detach().cpu(), a) <= 5e-5
