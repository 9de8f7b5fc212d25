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


def test_captured_graph_survives_cache_eviction_and_table_growth(monkeypatch):
    """A captured step holds raw pointers to cached tile / segment maps and to the positional table.  Evicting those
    cache entries (many other ragged shapes) and re-allocating the table larger must not change what a replay computes:
    the graph entry keeps its own references (lifetime.py).  The freed-and-reused case is provoked by filling new
    allocations with garbage before the replay."""
    from mil_amd.bags import BagLayout
    from mil_amd.segments import AttnSegs
    monkeypatch.setattr(BagLayout, "CACHE_ENTRIES", 4)
    monkeypatch.setattr(AttnSegs, "CACHE_ENTRIES", 4)
    BagLayout._cache.clear()
    AttnSegs._cache.clear()
    m = _model(0, layers=1)
    crit = torch.nn.BCELoss()
    params = [p for p in m.parameters() if p.requires_grad]
    gs = GraphedStep(params, max_graphs=1)
    lengths = [40, 64]

    def run(seed, graphed):
        x = syn.make_bags(seed, 2, 64, 768).to(DEV)
        t = m.clinic_extractor(syn.make_token_ids(seed, 2, 1).to(DEV))
        y = syn.make_labels(seed, 2).to(DEV)
        def body(x_, t_, y_):
            prob = m([x_], None, lengths, text_features=t_)[0]
            return crit(prob, y_), prob
        if graphed:
            out = gs.run(tuple(lengths), (x, t, y), body)
            return float(out[0]), [None if p.grad is None else p.grad.clone() for p in params]
        for p in params:
            p.grad = None
        loss = body(x, t, y)[0]
        loss.backward()
        return float(loss.detach()), [None if p.grad is None else p.grad.clone() for p in params]

    run(1, True)                      # first sight: eager
    run(2, True)                      # second: captured + replayed
    assert gs.replays == 1
    # other shapes, eagerly: evict every cache entry the graph used; one bag longer than the table: it is re-allocated
    for i, n in enumerate([33, 47, 52, 61, 38, 45, 2500]):
        x = syn.make_bags(50 + i, 1, n, 768).to(DEV)
        t = m.clinic_extractor(syn.make_token_ids(i, 1, 1).to(DEV))
        crit(m([x], None, None, text_features=t)[0], syn.make_labels(i, 1).to(DEV)).backward()
    assert len(BagLayout._cache) <= 4 and len(AttnSegs._cache) <= 4
    junk = [torch.full((4096,), 0x7fffffff, device=DEV, dtype=torch.int32) for _ in range(256)]   # reuse freed blocks
    torch.cuda.synchronize()
    l_g, g_g = run(3, True)
    assert gs.replays == 2
    l_e, g_e = run(3, False)
    del junk
    assert abs(l_g - l_e) <= 1e-6
    assert sum(g is not None for g in g_e) > 10
    for a, b in zip(g_g, g_e):
        assert (a is None) == (b is None)
        if b is not None:
            assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max()))
