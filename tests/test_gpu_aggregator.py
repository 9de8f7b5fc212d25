"""GPU parity of the whole drop-in module `aggregator(args).forward(x_list, x_CI)` (pathology + clinical text
branch, BASELINE config 3) against the golden vectors made by wiring the reference's leaf modules as
model/aggregator.py:134-209 does, and of the image-only variant (aggregator_clip.py)."""
from types import SimpleNamespace

import pytest
import torch

from conftest import check_grad, load_golden, rel_err
from mil_amd import synthetic as syn
from mil_amd.model.utils import get_model
from mil_amd.model.utils_clip import get_model as get_model_clip
from oracle import mil_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def make_args(**kw):
    a = dict(modality=["pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL", num_classes=2,
             learnablePrompt=0, alignment_base="CI", model_CT="resnetMC3_18")
    a.update(kw)
    return SimpleNamespace(**a)


@pytest.mark.parametrize("tag", ["fused_small_clip", "fused_P10", "fused_vitb32"])
def test_fused_module_vs_golden(tag):
    g = load_golden(tag)
    seed = int(g["seed"])
    B, N, P, clayers, cwidth, cvocab, cheads = [int(v) for v in g["cfg"]]
    p = syn.fused_params(seed, "TwoWayTransformer_Pth", clip_width=cwidth, clip_layers=clayers, clip_vocab=cvocab)
    model = get_model(make_args(clip_layers=clayers, clip_width=cwidth, clip_vocab=cvocab, clip_heads=cheads))
    missing, unexpected = model.load_state_dict(p, strict=False)
    assert not unexpected, unexpected
    model = model.to(DEV).eval()
    x = syn.make_bags(seed + 3, B, N, 768).to(DEV)
    ids = syn.make_token_ids(seed + 4, B, P, vocab=cvocab).to(DEV)
    y = syn.make_labels(seed + 5, B).to(DEV)
    prob, q = model([x], ids)
    loss = torch.nn.BCELoss()(prob, y)
    loss.backward()
    assert float((model.last_logits.detach().cpu() - g["logits"]).abs().max()) <= 2e-5         # bar: 1e-3
    assert torch.equal(prob.detach().cpu().argmax(-1), g["prob"].argmax(-1))                   # top-1 bit-exact
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5
    assert rel_err(q.detach().cpu(), g["x_Pth2CI"]) <= 5e-5
    live = [k for k in p if not k.startswith("clinic_extractor.")]
    params = dict(model.named_parameters())
    for k in live:
        gn = float(g["g." + k + ".norm"])
        got = params[k].grad if params[k].grad is not None else torch.zeros_like(params[k])
        if gn == 0.0:
            assert float(got.abs().max()) <= 1e-10, k
        else:
            check_grad("g." + k, got, g, 1e-3)                                                 # bar: rel 1e-3
    # frozen text tower: no gradient
    assert all(v.grad is None for k, v in params.items() if k.startswith("clinic_extractor."))


def test_state_dict_keys_match_reference_schema():
    model = get_model(make_args(clip_layers=1))
    keys = set(model.state_dict().keys())
    for k in ["aggregator.attention_V.0.weight", "aggregator.attention_U.0.bias", "aggregator.attention_weights.weight",
              "fc.1.weight", "fc_pathology.0.weight", "fc_CI2CT.0.weight", "fc_CI2Pth.0.bias", "fc_CI.0.weight",
              "prompt_embedding", "TwoWayTransformer_Pth.layers.0.self_attn.q_proj.weight",
              "TwoWayTransformer_Both.layers.1.cross_attn_image_to_token.out_proj.bias",
              "TwoWayTransformer_Pth.layers.1.mlp.lin1.weight", "TwoWayTransformer_Pth.norm_final_attn.weight",
              "TwoWayTransformer_Pth.final_attn_token_to_image.k_proj.weight",
              "clinic_extractor.model.token_embedding.weight", "clinic_extractor.model.positional_embedding",
              "clinic_extractor.model.transformer.resblocks.0.attn.in_proj_weight",
              "clinic_extractor.model.transformer.resblocks.0.mlp.c_fc.weight", "clinic_extractor.model.ln_final.weight",
              "clinic_extractor.model.text_projection", "clinic_extractor.model.logit_scale",
              "extractor_pathology.attention_V.0.weight"]:
        assert k in keys, k
    sd = model.state_dict()
    assert tuple(sd["TwoWayTransformer_Pth.layers.0.cross_attn_token_to_image.q_proj.weight"].shape) == (256, 512)
    assert tuple(sd["TwoWayTransformer_Pth.layers.0.self_attn.q_proj.weight"].shape) == (512, 512)
    assert tuple(sd["fc_pathology.0.weight"].shape) == (512, 768)
    assert "pe" not in keys and "_pe" not in keys          # plain attribute upstream too (aggregator.py:101-106)


def test_padded_batch_with_lengths_equals_per_bag():
    """dataset.py:386-391 zero-pads bags to a common length when batch > 1; with `lengths` the padding is dropped."""
    model = get_model(make_args(clip_layers=1)).to(DEV).eval()
    ns = [40, 100]
    x = torch.zeros(2, 100, 768)
    gen = torch.Generator().manual_seed(4)
    for b, n in enumerate(ns):
        x[b, :n] = torch.randn(n, 768, generator=gen)
    ids = syn.make_token_ids(8, 2, 1).to(DEV)
    with torch.no_grad():
        pb, _ = model([x.to(DEV)], ids, lengths=ns)
        for b, n in enumerate(ns):
            p1, _ = model([x[b:b + 1, :n].to(DEV)], ids[b:b + 1])
            assert float((pb[b] - p1[0]).abs().max()) <= 1e-6


def test_ci_only_branch_runs():
    model = get_model(make_args(modality=["CI"], clip_layers=1)).to(DEV).eval()
    ids = syn.make_token_ids(9, 3, 10).to(DEV)
    with torch.no_grad():
        out = model([], ids)
    assert tuple(out.shape) == (3, 2) and bool(((out > 0) & (out < 1)).all())


def test_out_of_scope_branches_raise():
    m = get_model(make_args(modality=["CT", "pathology"], clip_layers=1))       # built since round 2 (precomputed CT map)
    with pytest.raises(NotImplementedError):                                      # ... but the CT encoder itself is not
        m.to(DEV)([torch.randn(1, 512, 4, 2, 2, device=DEV, requires_grad=True), torch.randn(1, 8, 768, device=DEV)],
                  syn.make_token_ids(1, 1, 1).to(DEV))
    with pytest.raises(NotImplementedError):
        get_model(make_args(aggregator="TransMIL"))
    with pytest.raises(NotImplementedError):
        get_model(make_args(model_CI="simpleFCs_v1"))


def test_image_only_module_vs_oracle():
    args = make_args(patch_dim=512)
    m = get_model_clip(args)
    p = syn.image_only_params(21, L=512)
    m.load_state_dict({k.replace("aggregator.", "extractor_pathology."): v for k, v in p.items()})
    m = m.to(DEV).eval()
    x = syn.make_bags(5, 3, 77, 512)
    emb, prob = m([x.to(DEV)])
    for b in range(3):
        o = orc.image_only_forward(x[b], p)
        assert float((m.last_logits[b].cpu() - o["logits"][0]).abs().max()) <= 2e-5
        assert float((emb[b].detach().cpu() - o["M"][0]).abs().max()) <= 1e-5


def test_ten_prompts_ragged_batch_absorbed_path_matches_the_general_path(monkeypatch):
    """10 prompts per bag, ragged bags: the multi-token absorbed attention (grouped skinny products, two-segment bag layout)
    against the general projection + attention-core path of the same module - outputs and every parameter gradient -
    and against the same bags run one at a time."""
    from mil_amd import ops
    torch.manual_seed(11)
    model = get_model(make_args(clip_layers=1)).to(DEV).eval()
    ns = [40, 100, 77]
    x = torch.zeros(3, 100, 768)
    gen = torch.Generator().manual_seed(5)
    for b, n in enumerate(ns):
        x[b, :n] = torch.randn(n, 768, generator=gen)
    ids = syn.make_token_ids(12, 3, 10).to(DEV)
    y = syn.make_labels(13, 3).to(DEV)

    def run():
        model.zero_grad()
        prob, _ = model([x.to(DEV)], ids, lengths=ns)
        torch.nn.BCELoss()(prob, y).backward()
        return prob.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    p_abs, g_abs = run()
    monkeypatch.setattr(ops, "multi_token_ok", lambda *a, **k: False)
    p_gen, g_gen = run()
    assert float((p_abs - p_gen).abs().max()) <= 2e-6
    # token->image k_proj.bias is softmax-invariant: exactly no gradient on the absorbed path, rounding noise on the general one
    assert {k for k in g_gen if not k.endswith("k_proj.bias")} <= set(g_abs)
    for k in g_gen:
        if float(g_gen[k].norm()) > 1e-7 and not k.endswith("k_proj.bias"):
            assert rel_err(g_abs[k].cpu(), g_gen[k].cpu()) <= 2e-4, k
    monkeypatch.undo()
    with torch.no_grad():
        for b, n in enumerate(ns):
            p1, _ = model([x[b:b + 1, :n].to(DEV)], ids[b:b + 1])
            assert float((p_abs[b] - p1[0]).abs().max()) <= 2e-6


@pytest.mark.parametrize("num_classes,ns", [(2, [40, 100, 1]), (3, [64, 64])])
def test_fused_pool_head_loss_node_matches_the_op_by_op_tail(num_classes, ns):
    """forward(labels=y): pool + head + loss as one autograd node (ops.gated_pool_head_loss) against the same module run op by
    op with the criterion outside (train_ddp.py:95-99,323-324) - loss, outputs and every parameter gradient."""
    torch.manual_seed(5)
    model = get_model(make_args(clip_layers=1, num_classes=num_classes)).to(DEV).eval()
    B, N = len(ns), max(ns)
    x = torch.zeros(B, N, 768)
    gen = torch.Generator().manual_seed(4)
    for b, n in enumerate(ns):
        x[b, :n] = torch.randn(n, 768, generator=gen)
    x = x.to(DEV)
    ids = syn.make_token_ids(8, B, 1).to(DEV)
    y = torch.zeros(B, num_classes, device=DEV)
    y[torch.arange(B), torch.arange(B) % num_classes] = 1.0
    crit = torch.nn.CrossEntropyLoss() if num_classes > 2 else torch.nn.BCELoss()
    prob, _ = model([x], ids, lengths=ns)
    loss_ref = crit(prob, y)
    loss_ref.backward()
    ref = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    z_ref = model.last_logits.detach().clone()
    model.zero_grad(set_to_none=True)
    prob2, _ = model([x], ids, lengths=ns, labels=y)
    assert model.last_loss is not None and model.last_loss.requires_grad
    model.last_loss.backward()
    assert abs(float(model.last_loss) - float(loss_ref)) <= 1e-6 * max(1.0, abs(float(loss_ref)))
    assert float((prob2 - prob.detach()).abs().max()) <= 1e-6
    assert float((model.last_logits - z_ref).abs().max()) <= 1e-5
    got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert set(got) == set(ref)
    for k in ref:
        if k.endswith("k_proj.bias") or float(ref[k].norm()) < 1e-7:
            # mathematically-zero gradients (softmax is invariant to a constant on every score): rounding noise on both sides
            assert float(got[k].abs().max()) < 5e-5, k
        else:
            assert rel_err(got[k], ref[k]) <= 2e-4, (k, rel_err(got[k], ref[k]))
