"""GPU: hipGraph replay of the autograd training step (graph_step.GraphedStep) and the fixed-shape text tower it needs
(CLIPText.static_rows): replayed steps must reproduce the eager ones."""
import copy
from types import SimpleNamespace

import pytest
import torch

from conftest import rel_err
from mil_amd import synthetic as syn
from mil_amd.graph_step import GraphedStep
from mil_amd.model.utils import get_model
from mil_amd.optim import FlatAdam, FlatSGD

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(learnable, layers=2):
    args = SimpleNamespace(modality=["pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL",
                           num_classes=2, learnablePrompt=learnable, n_ctx=4, clinical_features=["a", "b"], clip_layers=layers)
    torch.manual_seed(11)
    return get_model(args).to(DEV).eval()          # eval: no dropout, so eager and replayed steps are comparable


def test_static_tower_matches_live_prefix_tower():
    m = _model(1)
    ids = syn.make_token_ids(4, 2, 3).to(DEV)
    ext = m.clinic_extractor
    go = torch.randn((2, 3, 512), generator=torch.Generator().manual_seed(1)).to(DEV)
    res = []
    for static in (False, True):
        ext.model.static_rows = static
        ext.ctx.grad = None
        out = ext(ids)
        (out * go).sum().backward()
        res.append((out.detach().clone(), ext.ctx.grad.clone()))
    assert rel_err(res[1][0].cpu(), res[0][0].cpu()) <= 1e-5
    assert rel_err(res[1][1].cpu(), res[0][1].cpu()) <= 1e-4
    with torch.no_grad():
        ext.model.static_rows = False
        a = ext.model.encode_text(ids.reshape(6, -1))
        ext.model.static_rows = True
        b = ext.model.encode_text(ids.reshape(6, -1))
    assert rel_err(b.cpu(), a.cpu()) <= 1e-5


@pytest.mark.parametrize("learnable", [0, 1])
def test_replayed_steps_reproduce_eager_training(learnable):
    ref = _model(learnable)
    ours = copy.deepcopy(ref)
    if learnable:
        ours.clinic_extractor.model.static_rows = True
    P = 3 if learnable else 1
    crit = torch.nn.BCELoss()
    mk = (lambda m: FlatSGD([p for p in m.parameters() if p.requires_grad], lr=1e-2)) if learnable else \
         (lambda m: FlatAdam([p for p in m.parameters() if p.requires_grad], lr=1e-4))
    o_ref, o_our = mk(ref), mk(ours)
    gs = GraphedStep(o_our.params)
    lengths = [50, 64]
    l_ref, l_our = [], []
    for step in range(6):
        x = syn.make_bags(100 + step, 2, 64, 768).to(DEV)
        ids = syn.make_token_ids(200 + step, 2, P).to(DEV)
        y = syn.make_labels(300 + step, 2).to(DEV)
        o_ref.zero_grad()
        loss = crit(ref([x], ids, lengths)[0], y)
        loss.backward()
        o_ref.step()
        l_ref.append(float(loss.detach()))
        if learnable:
            def body(x_, ids_, y_):
                prob = ours([x_], ids_, lengths)[0]
                return crit(prob, y_), prob
            out = gs.run(tuple(lengths), (x, ids, y), body)
        else:
            t = ours.clinic_extractor(ids)
            def body(x_, t_, y_):
                prob = ours([x_], None, lengths, text_features=t_)[0]
                return crit(prob, y_), prob
            out = gs.run(tuple(lengths), (x, t, y), body)
        o_our.step()
        l_our.append(float(out[0]))
    assert gs.eager_steps == 1 and gs.replays == 5
    assert max(abs(a - b) for a, b in zip(l_ref, l_our)) <= 2e-5, (l_ref, l_our)
    for (k, a), b in zip(ref.named_parameters(), ours.parameters()):
        if a.requires_grad:
            assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max())), k


def test_new_shape_runs_eagerly_then_gets_its_own_graph():
    m = _model(0, layers=1)
    opt = FlatAdam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
    gs = GraphedStep(opt.params, max_graphs=2)
    crit = torch.nn.BCELoss()
    for step, n in enumerate([32, 32, 48, 32, 48, 48]):
        x = syn.make_bags(step, 1, n, 768).to(DEV)
        t = m.clinic_extractor(syn.make_token_ids(step, 1, 1).to(DEV))
        y = syn.make_labels(step, 1).to(DEV)
        def body(x_, t_, y_):
            prob = m([x_], None, None, text_features=t_)[0]
            return crit(prob, y_), prob
        loss, prob = gs.run(n, (x, t, y), body)
        opt.step()
        assert torch.isfinite(loss).all() and tuple(prob.shape) == (1, 2)
    assert gs.eager_steps == 2 and gs.replays == 4
