"""GPU: the fusion model (`aggregator(args)`, pathology + one clinical note per bag) in the authors' regime - ONE ragged bag
per step whose length changes every step (reference run_train.sh:81; dataset.py:366-393).  Bag lengths live on the device
(segments.FusionBucket), the step of a capacity bucket is one captured hipGraph (fusion_step.RaggedFusionStepper): a stream
of 50 bags with 2 000 .. 15 592 patches must replay at most 8 graphs and match the oracle (orc.fused_forward) bag by bag."""
import copy
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import rel_err
from mil_amd import ops, synthetic as syn
from mil_amd.bags import BagLayout, bucket_rows
from mil_amd.fusion_step import RaggedFusionStepper
from mil_amd.model.utils import get_model
from mil_amd.optim import FlatAdam
from mil_amd.segments import AttnSegs, FusionBucket
from oracle import mil_oracle as orc

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _model(layers=2, seed=11):
    args = SimpleNamespace(modality=["pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL",
                           num_classes=2, learnablePrompt=0, n_ctx=4, clinical_features=["a", "b"], clip_layers=layers,
                           cache_text=0)
    torch.manual_seed(seed)
    return get_model(args).to(DEV).eval()          # eval: no dropout, comparable with the oracle


@pytest.mark.parametrize("lengths", [[2500], [300, 1, 77, 1500, 33], [3072], [1]])
def test_device_built_segments_equal_the_host_ones(lengths):
    B, N = len(lengths), sum(lengths)
    cap = bucket_rows(N)
    bk = FusionBucket(cap, B, DEV)
    bk._min_rows = 1                                # this test only compares the maps
    bk.set_lengths(lengths).refresh()
    torch.cuda.synchronize()
    ref = AttnSegs.make([1] * B, lengths, DEV)
    assert int(bk.rows_dev.item()) == N
    assert torch.equal(bk.k_off.cpu(), ref.k_off.cpu())
    assert torch.equal(bk.k_bag[:N].cpu(), ref.k_bag.cpu()) and bool((bk.k_bag[N:] == B - 1).all())
    assert torch.equal(bk.bag_tile64_off.cpu(), ref.bag_tile_off.cpu())
    assert torch.equal(bk.tile64[:ref.ntiles].cpu(), ref.tile_map.cpu())
    pad = bk.tile64[ref.ntiles:].cpu()                  # padding tiles {0, row0, -count}: exactly the rows [N, cap), in order
    covered = []
    for bag, row0, cnt in pad.tolist():
        assert bag == 0 and cnt <= 0
        covered += list(range(row0, row0 - cnt))
    assert covered == list(range(N, cap))
    lay = BagLayout.two_segment(lengths, [1] * B, DEV)
    tm = lay.tile_map.cpu().clone()
    tok = tm[:, 1] >= N                              # host layout puts the token rows right behind the N patch rows ...
    tm[tok, 1] += cap - N                            # ... the bucket keeps them behind its `cap` rows
    assert torch.equal(bk.bag_tile32_off.cpu(), lay.bag_tile_off.cpu())
    assert torch.equal(bk.tile32[:lay.T].cpu(), tm) and int(bk.tile32[lay.T:].abs().sum()) == 0
    rb = bk.row_bag_dev.cpu()
    assert torch.equal(rb[:N], ref.k_bag.cpu()) and bool((rb[N:cap] == -1).all())
    assert torch.equal(rb[cap:], torch.arange(B, dtype=torch.int32))


@pytest.mark.parametrize("lengths,P", [([2500], 1), ([1100, 1300], 1), ([2500], 10), ([700, 900], 3)])
def test_bucket_step_equals_the_exact_shape_step(lengths, P):
    """Same bags through aggregator.forward(lengths=...) (host-built maps, exact shapes) and through the bucket form: logits,
    token outputs and every parameter gradient must agree - padding rows have zero weight and zero gradient.  P = 1: the
    absorbed one-token kernels; P = 10 / 3: the multi-token grouped products (`CI_prompt_version='devided'`)."""
    m = _model()
    B, N = len(lengths), sum(lengths)
    cap = bucket_rows(N)
    gen = torch.Generator().manual_seed(5)
    bags = [torch.randn((n, 768), generator=gen) for n in lengths]
    ids = syn.make_token_ids(6, B, P).to(DEV)
    y = syn.make_labels(7, B).to(DEV)
    with torch.no_grad():
        t = m.clinic_extractor(ids)
    pad = torch.zeros((B, max(lengths), 768))
    for b, xb in enumerate(bags):
        pad[b, :xb.shape[0]] = xb
    m.zero_grad()
    prob_a, q_a = m([pad.to(DEV)], None, lengths, text_features=t, labels=y)
    za = m.last_logits.detach().clone()
    m.last_loss.backward()
    ga = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    bk = FusionBucket(cap, B, DEV, P).set_lengths(lengths)
    x = torch.full((cap, 768), 3.0, device=DEV)      # stale rows behind the bags must not matter
    x[:N] = torch.cat(bags, 0).to(DEV)
    m.zero_grad()
    prob_b, q_b = m([x], None, text_features=t, labels=y, bucket=bk)
    zb = m.last_logits.detach().clone()
    m.last_loss.backward()
    torch.cuda.synchronize()
    assert float((za - zb).abs().max()) <= 1e-6 and torch.equal(prob_a.argmax(-1), prob_b.argmax(-1))
    assert rel_err(q_b.detach().cpu(), q_a.detach().cpu()) <= 1e-6
    gb = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert set(ga) == set(gb)
    for k in ga:
        if float(ga[k].norm()) < 1e-7:               # mathematically zero (softmax bias): rounding noise on both sides
            assert float(gb[k].abs().max()) < 1e-6, k
        else:
            # P > 1: the grouped products are launched for `cap` rows per group instead of the longest bag - other split
            # counts and tile shapes, i.e. another summation order (4e-5 on the smallest gradients; the module bar is 1e-3)
            tol = 2e-5 if P == 1 else 2e-4
            assert rel_err(gb[k].cpu(), ga[k].cpu()) <= tol, (k, rel_err(gb[k].cpu(), ga[k].cpu()))


def test_stream_of_ragged_fusion_bags_replays_few_graphs_and_matches_the_oracle():
    m = _model(layers=1)
    lr = 1e-5
    opt = FlatAdam([p for p in m.parameters() if p.requires_grad], lr=lr, weight_decay=1e-7, counted=True)
    st = RaggedFusionStepper(m, opt, B=1)
    rng = np.random.default_rng(5)
    names = [k for k, p in m.named_parameters() if p.requires_grad]
    frozen = {k: v.detach().cpu() for k, v in m.state_dict().items() if k not in names}
    worst = 0.0
    for step in range(50):
        n = int(rng.integers(2000, 15593))
        x = torch.randn((n, 768), generator=torch.Generator().manual_seed(1000 + step))
        ids = syn.make_token_ids(2000 + step, 1, 1)
        y = syn.make_labels(3000 + step, 1)
        slot = st.slot(n)
        slot.x[:n].copy_(x.to(DEV))
        slot.y.copy_(y.to(DEV))
        st.encode_notes(slot, ids.to(DEV))
        # the oracle on the weights this step starts from (the fused update overwrites them)
        sd = dict(frozen)
        sd.update({k: p.detach().cpu().clone() for k, p in m.named_parameters() if p.requires_grad})
        loss, prob, z = st.step(slot, [n])
        torch.cuda.synchronize()
        if step % 3 == 0 or n > 12000:               # every third bag and every long one against the CPU oracle
            with torch.no_grad():
                o = orc.fused_forward(x, ids[0], sd)
            d = float((z.cpu() - o["logits"]).abs().max())
            worst = max(worst, d)
            assert d <= 1e-3, (step, n, d)                                  # north_star bar; measured ~1e-6
            assert torch.equal(prob.cpu().argmax(-1), o["prob"].argmax(-1)), (step, n)
            assert abs(float(loss) - float(orc.bce_loss(o["prob"], y))) <= 1e-4
    assert len(st.slots) <= 8 and len(st.gs._graphs) <= 8
    assert st.replays >= 50 - 2 * len(st.slots) and st.replays + st.eager_steps == 50
    assert int(opt.step_counter.item()) == 50                               # every step updated the parameters once
    print(f"fusion stream: {len(st.slots)} buckets, {st.replays} replays, max |dlogit| vs oracle {worst:.2e}")


def test_bucketed_fusion_training_tracks_the_exact_shape_training():
    """Optimizer inside the graph: 8 steps over two buckets with a learning-rate change half way must leave the same
    parameters as eager exact-shape steps with the same FlatAdam arithmetic."""
    ref = _model(layers=1)
    ours = copy.deepcopy(ref)
    mk = lambda mm, c: FlatAdam([p for p in mm.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-7, counted=c)   # noqa: E731
    o_ref, o_our = mk(ref, False), mk(ours, True)
    st = RaggedFusionStepper(ours, o_our, B=1)
    lengths = [2100, 2900, 2000, 3000, 1900, 2800, 2040, 3050]
    for step, n in enumerate(lengths):
        lr = 1e-4 if step < 4 else 2e-6
        o_ref.param_groups[0]["lr"] = lr
        o_our.param_groups[0]["lr"] = lr
        x = torch.randn((n, 768), generator=torch.Generator().manual_seed(50 + step)).to(DEV)
        ids = syn.make_token_ids(60 + step, 1, 1).to(DEV)
        y = syn.make_labels(70 + step, 1).to(DEV)
        with torch.no_grad():
            t = ref.clinic_extractor(ids)
        o_ref.zero_grad()
        ref([x.unsqueeze(0)], None, text_features=t, labels=y)
        ref.last_loss.backward()
        o_ref.step()
        slot = st.slot(n)
        slot.x[:n].copy_(x)
        slot.y.copy_(y)
        st.encode_notes(slot, ids)
        loss, _, _ = st.step(slot, [n])
        assert abs(float(loss) - float(ref.last_loss.detach())) <= 2e-6, step
    torch.cuda.synchronize()
    assert len(st.slots) == 2 and st.replays == 6              # per bucket: one eager visit, then capture + replays
    start = dict(_model(layers=1).named_parameters())
    pr, po = dict(ref.named_parameters()), dict(ours.named_parameters())
    for k, p0 in start.items():
        if not p0.requires_grad or k.endswith("attention_weights.bias"):
            continue          # softmax bias: a mathematically zero gradient, Adam normalises its rounding noise to +-lr
        moved = float((pr[k].detach() - p0.detach()).abs().max())
        assert float((po[k].detach() - pr[k].detach()).abs().max()) <= 0.02 * moved + 1e-9, (k, moved)


def test_bucket_step_with_learnable_prompts_equals_the_exact_shape_step():
    """Upstream's default `--learnablePrompt 1`: P = len(clinical_features) + 1 prompts per bag, the context vectors trained
    THROUGH the frozen text tower (model/dim1/CLIP.py:29-62).  Bucket form vs exact shapes: loss, logits, d ctx and the
    fusion parameters' gradients."""
    args = SimpleNamespace(modality=["pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL", num_classes=2,
                           learnablePrompt=1, n_ctx=4, clinical_features=["a", "b"], clip_layers=2, cache_text=0)
    torch.manual_seed(13)
    m = get_model(args).to(DEV).eval()
    n, P = 2300, 3
    cap = bucket_rows(n)
    xb = torch.randn((n, 768), generator=torch.Generator().manual_seed(8))
    ids = syn.make_token_ids(9, 1, P).to(DEV)
    y = syn.make_labels(10, 1).to(DEV)
    m.zero_grad()
    m([xb.unsqueeze(0).to(DEV)], ids, labels=y)
    za, la = m.last_logits.detach().clone(), float(m.last_loss.detach())
    m.last_loss.backward()
    ga = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    bk = FusionBucket(cap, 1, DEV, P).set_lengths([n])
    x = torch.full((cap, 768), -2.0, device=DEV)
    x[:n] = xb.to(DEV)
    m.zero_grad()
    m([x], ids, labels=y, bucket=bk)
    zb, lb = m.last_logits.detach().clone(), float(m.last_loss.detach())
    m.last_loss.backward()
    torch.cuda.synchronize()
    assert float((za - zb).abs().max()) <= 1e-6 and abs(la - lb) <= 1e-6
    gb = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert "clinic_extractor.ctx" in ga and set(ga) == set(gb)
    for k in ga:
        if float(ga[k].norm()) < 1e-7:
            assert float(gb[k].abs().max()) < 1e-6, k
        else:
            assert rel_err(gb[k].cpu(), ga[k].cpu()) <= 2e-4, (k, rel_err(gb[k].cpu(), ga[k].cpu()))


def test_ragged_stream_with_ten_prompts_and_with_learnable_prompts_replays():
    """The stepper beyond one note per bag: 10 frozen prompts (Adam inside the graph) and learnable prompts (the tower inside
    the step, flat SGD outside): two buckets, each captured on its second visit, losses equal to eager exact-shape steps of a
    twin model."""
    from mil_amd.optim import FlatSGD
    for learnable in (0, 1):
        P = 3 if learnable else 10
        args = SimpleNamespace(modality=["pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL",
                               num_classes=2, learnablePrompt=learnable, n_ctx=4, clinical_features=["a", "b"], clip_layers=1,
                               cache_text=0)
        torch.manual_seed(21)
        ref = get_model(args).to(DEV).eval()
        ours = copy.deepcopy(ref)
        mk = (lambda mm, c: FlatSGD([p for p in mm.parameters() if p.requires_grad], lr=1e-3)) if learnable else \
             (lambda mm, c: FlatAdam([p for p in mm.parameters() if p.requires_grad], lr=1e-4, counted=c))
        o_ref, o_our = mk(ref, False), mk(ours, True)
        st = RaggedFusionStepper(ours, o_our, B=1, P=P, learnable=bool(learnable), opt_in_graph=not learnable)
        for step, n in enumerate([2100, 2900, 2000, 3000, 1900, 2800]):
            x = torch.randn((n, 768), generator=torch.Generator().manual_seed(150 + step)).to(DEV)
            ids = syn.make_token_ids(160 + step, 1, P).to(DEV)
            y = syn.make_labels(170 + step, 1).to(DEV)
            o_ref.zero_grad()
            if learnable:
                ref([x.unsqueeze(0)], ids, labels=y)
            else:
                with torch.no_grad():
                    t = ref.clinic_extractor(ids)
                ref([x.unsqueeze(0)], None, text_features=t, labels=y)
            ref.last_loss.backward()
            o_ref.step()
            slot = st.slot(n)
            slot.x[:n].copy_(x)
            slot.y.copy_(y)
            if learnable:
                slot.ids.copy_(ids)
            else:
                st.encode_notes(slot, ids)
            loss, _, _ = st.step(slot, [n])
            assert abs(float(loss) - float(ref.last_loss.detach())) <= 5e-6, (learnable, step)
        assert len(st.slots) == 2 and st.replays == 4, (learnable, st.replays)


def test_multi_tail_layout_equals_the_host_multi_segment_layout():
    """The CT + pathology bag (aggregator.py:173): bucket tail [P, D, P] against BagLayout.multi_segment."""
    lengths, P, D = [300, 77, 1500], 1, 160
    B, N = len(lengths), sum(lengths)
    cap = bucket_rows(N)
    bk = FusionBucket(cap, B, DEV, P, tail=[P, D, P])
    bk._min_rows = 1
    bk.set_lengths(lengths).refresh()
    torch.cuda.synchronize()
    lay = BagLayout.multi_segment([lengths, [P] * B, [D] * B, [P] * B], DEV)
    tm = lay.tile_map.cpu().clone()
    tail = tm[:, 1] >= N
    tm[tail, 1] += cap - N
    assert bk.layout.R == cap + B * (2 * P + D)
    assert torch.equal(bk.bag_tile32_off.cpu(), lay.bag_tile_off.cpu())
    assert torch.equal(bk.tile32[:lay.T].cpu(), tm) and int(bk.tile32[lay.T:].abs().sum()) == 0
    rb = bk.row_bag_dev.cpu()
    want = torch.cat([torch.arange(B).repeat_interleave(P), torch.arange(B).repeat_interleave(D),
                      torch.arange(B).repeat_interleave(P)]).to(torch.int32)
    assert bool((rb[N:cap] == -1).all()) and torch.equal(rb[cap:], want)


def test_ct_plus_pathology_bucket_step_equals_the_exact_shape_step():
    """The authors' own run (run_train.sh:81: CT + pathology, one note, frozen tower, loss_point CT-Pth-Last) in bucket form
    vs exact shapes: logits, both text-aligned tokens, the cosine term and every gradient - TwoWayTransformer_Both is used
    twice per forward, once on static CT tokens and once on the device-length patch rows."""
    args = SimpleNamespace(modality=["CT", "pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL",
                           num_classes=2, learnablePrompt=0, alignment_base="CI", model_CT="resnetMC3_18", clip_layers=1,
                           cache_text=0)
    torch.manual_seed(17)
    m = get_model(args).to(DEV).eval()
    lengths, D, hw = [1200, 900], 160, 2
    B, N = len(lengths), sum(lengths)
    cap = bucket_rows(N)
    gen = torch.Generator().manual_seed(19)
    bags = [torch.randn((n, 768), generator=gen) for n in lengths]
    ids = syn.make_token_ids(20, B, 1).to(DEV)
    y = syn.make_labels(21, B).to(DEV)
    ct = syn.make_ct_map(22, B, D, hw).to(DEV)
    with torch.no_grad():
        t = m.clinic_extractor(ids)
    pad = torch.zeros((B, max(lengths), 768))
    for b, xb in enumerate(bags):
        pad[b, :xb.shape[0]] = xb
    scale = 3.0 / (B * 2)                                                    # loss_point CT-Pth-Last: three terms, one head

    def run(**kw):
        m.zero_grad()
        prob, q_ct, q_p = m(kw.pop("xs"), None, text_features=t, labels=y, loss_scale=scale, **kw)
        cs = ops.cosine_embedding_loss(q_ct.squeeze(1), q_p.squeeze(1))
        (m.last_loss + cs).backward()
        return (m.last_logits.detach().clone(), q_ct.detach().clone(), q_p.detach().clone(), float(cs.detach()),
                {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})

    za, qa, pa, ca, ga = run(xs=[ct, pad.to(DEV)], lengths=lengths)
    bk = FusionBucket(cap, B, DEV, 1, tail=[1, D, 1]).set_lengths(lengths)
    x = torch.full((cap, 768), 1.5, device=DEV)
    x[:N] = torch.cat(bags, 0).to(DEV)
    zb, qb, pb, cb, gb = run(xs=[ct, x], bucket=bk)
    torch.cuda.synchronize()
    assert float((za - zb).abs().max()) <= 1e-6 and abs(ca - cb) <= 1e-6
    assert rel_err(qb.cpu(), qa.cpu()) <= 1e-6 and rel_err(pb.cpu(), pa.cpu()) <= 1e-6
    assert set(ga) == set(gb)
    for k in ga:
        if float(ga[k].norm()) < 1e-7:
            assert float(gb[k].abs().max()) < 1e-6, k
        else:
            assert rel_err(gb[k].cpu(), ga[k].cpu()) <= 5e-5, (k, rel_err(gb[k].cpu(), ga[k].cpu()))


def test_bucketed_inference_replays_and_equals_the_exact_forward():
    """Evaluation (reference test_ddp.py:187-253: eval mode, one bag per forward): forward-only graphs per capacity bucket
    (fusion_step.RaggedFusionInference) against the exact-shape eager forward, bag by bag."""
    from mil_amd.fusion_step import RaggedFusionInference
    m = _model(layers=1)
    ref = copy.deepcopy(m)                              # the twin keeps the live-prefix text tower
    inf = RaggedFusionInference(m, B=1, P=1)            # text tower inside the replayed forward (fixed-shape form)
    for step, n in enumerate([2100, 2900, 2000, 3000, 1900, 2800, 2500]):
        x = torch.randn((n, 768), generator=torch.Generator().manual_seed(400 + step)).to(DEV)
        ids = syn.make_token_ids(410 + step, 1, 1).to(DEV)
        with torch.no_grad():
            want, _ = ref([x.unsqueeze(0)], ids)
        slot = inf.slot(n)
        slot.x[:n].copy_(x)
        slot.ids.copy_(ids)
        got = inf.forward(slot, [n])
        torch.cuda.synchronize()
        assert float((got - want).abs().max()) <= 2e-6, (step, n)
    assert len(inf.slots) == 2 and inf.replays == 5 and inf.eager == 2
