"""GPU: the HBM-resident cohort (cohort.DeviceCohort, csrc/cohort.hip) - the on-device form of the reference's input
pipeline (dataset.py:366-393 patch drop, train_ddp.py:193,274-293 loader).  The drawn subsets must equal the numpy
restatement bit for bit, the feed launch must place exactly those rows (and lengths / labels / notes) in the bucket, and a
training step fed this way must equal the oracle on the same rows (1e-3 / top-1, north_star)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from mil_amd import synthetic as syn
from mil_amd.cohort import DeviceCohort, HostFeed, keep_count
from mil_amd.trainer import ImageOnlyTrainer, RaggedImageOnlyStepper
from oracle import cohort as oc
from oracle import mil_oracle as orc

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _cohort(ns, F=64, keeps=None, seed=77, P=1, C=2):
    g = torch.Generator().manual_seed(3)
    bags = [torch.randn((n, F), generator=g) for n in ns]
    labels = syn.make_labels(4, len(ns), C)
    ids = syn.make_token_ids(5, len(ns), P)
    co = DeviceCohort(bags, labels, DEV, ids=ids, keep=keeps, seed=seed)
    return co, bags, labels, ids


def test_drawn_subsets_equal_the_oracle_bit_for_bit():
    ns = [1, 3, 4, 5, 10, 1023, 1024, 1025, 4097, 15592, 70001, 2]
    keeps = [0.9, 0.8, 0.9, 0.8, 0.9, 0.8, 0.9, 0.8, 0.9, 0.8, 0.9, 1.0]
    co, *_ = _cohort(ns, F=4, keeps=keeps)
    for epoch in (0, 1, 2 ** 33 + 5):
        co.draw_epoch(epoch)
        torch.cuda.synchronize()
        got = co.sel.cpu().numpy()[:co.sel_off[-1]]
        want = oc.select_epoch(np.asarray(co.row_off), co.k_train, co.seed, epoch)
        assert np.array_equal(got, want), epoch
        for j, n in enumerate(ns):                                 # the reference's contract, directly
            part = got[co.sel_off[j]:co.sel_off[j + 1]] - co.row_off[j]
            assert len(part) == int(n * keeps[j]) and (np.diff(part) > 0).all()
            assert len(part) == 0 or (part[0] >= 0 and part[-1] < n)


def test_feed_places_exactly_the_drawn_rows_lengths_labels_and_notes():
    ns = [300, 41, 1500, 97, 222, 64, 130, 77, 512, 33, 2048]
    keeps = [0.9, 0.8] * 5 + [0.9]
    co, bags, labels, ids = _cohort(ns, F=64, keeps=keeps, P=2)
    text = torch.randn((len(ns), 2, 512), generator=torch.Generator().manual_seed(9))
    co.set_text(text)
    co.draw_epoch(4)
    sel = oc.select_epoch(np.asarray(co.row_off), co.k_train, co.seed, 4)
    flat = torch.cat(bags, 0)
    for idxs in ([2], [0, 1, 3], list(range(11)), [10, 2]):           # 11 bags: two launches (8 + 3)
        B = len(idxs)
        x = torch.full((4096 + 1024, 64), 7.0, device=DEV)
        len_dev = torch.full((B,), -1, device=DEV, dtype=torch.int32)
        y = torch.zeros((B, 2), device=DEV)
        ids_d = torch.zeros((B, 2, 77), device=DEV, dtype=torch.int64)
        txt_d = torch.zeros((B, 2, 512), device=DEV)
        ks = co.feed(idxs, x, len_dev, y, ids_d, txt_d)
        torch.cuda.synchronize()
        assert ks == [int(ns[j] * keeps[j]) for j in idxs] and len_dev.cpu().tolist() == ks
        want = torch.cat([flat[sel[co.sel_off[j]:co.sel_off[j + 1]]] for j in idxs], 0)
        assert torch.equal(x[:sum(ks)].cpu(), want)
        assert bool((x[sum(ks):] == 7.0).all())                       # rows behind the bags are not touched
        assert torch.equal(y.cpu(), labels[idxs]) and torch.equal(ids_d.cpu(), ids[idxs]) and torch.equal(txt_d.cpu(), text[idxs])
    # no augmentation (validation / --augmentation 0): the whole bags
    co.draw_epoch(0, augment=False)
    x = torch.zeros((2048, 64), device=DEV)
    assert co.feed([1, 4], x) == [41, 222]
    assert torch.equal(x[:263].cpu(), torch.cat([bags[1], bags[4]], 0))


def test_host_feed_draws_and_places_the_same_rows_as_the_resident_cohort():
    ns = [300, 1500, 97, 2048, 5]
    keeps = [0.9, 0.8, 0.9, 0.8, 0.9]
    co, bags, labels, ids = _cohort(ns, F=64, keeps=keeps)
    co.draw_epoch(2)
    hf = HostFeed(lambda j: bags[j].numpy(), ns, 64, labels, DEV, ids=ids, keep=keeps, seed=co.seed)
    order = [3, 0, 1, 4, 2, 3, 3]
    hf.prefetch(order[0])
    for t, j in enumerate(order):
        a, b = torch.zeros((2048, 64), device=DEV), torch.zeros((2048, 64), device=DEV)
        la, lb = torch.zeros(1, device=DEV, dtype=torch.int32), torch.zeros(1, device=DEV, dtype=torch.int32)
        ya, yb = torch.zeros((1, 2), device=DEV), torch.zeros((1, 2), device=DEV)
        k = hf.next(a, la, ya, epoch=2)
        if t + 1 < len(order):
            hf.prefetch(order[t + 1])                    # the next bag loads while this one is consumed
        assert co.feed([j], b, lb, yb) == [k]
        torch.cuda.synchronize()
        assert torch.equal(a, b) and torch.equal(la, lb) and torch.equal(ya, yb)
    assert hf.next.__doc__                                # (documented entry)


def test_image_only_steps_fed_from_the_cohort_match_the_oracle_on_the_same_rows():
    """The authors' regime end to end on the device: per epoch one select launch, per step one feed launch + one replayed
    graph.  Eval-mode arithmetic (no dropout) so that the oracle + torch.optim.Adam can run in lock-step on the rows the
    numpy restatement says were drawn."""
    L, lr = 512, 1e-3
    rng = np.random.default_rng(8)
    ns = [int(v) for v in rng.integers(2000, 6000, size=12)]
    keeps = [0.9 if i % 3 else 0.8 for i in range(len(ns))]
    co, bags, labels, _ = _cohort(ns, F=L, keeps=keeps, seed=1234)
    p = syn.image_only_params(41, L=L)
    tr = ImageOnlyTrainer(p, DEV, lr=lr, counted=True)
    st = RaggedImageOnlyStepper(tr, B=1)
    ref = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    opt = torch.optim.Adam(list(ref.values()), lr=lr, betas=(0.9, 0.999), weight_decay=1e-7)
    for epoch in range(2):
        co.draw_epoch(epoch)
        for j in rng.permutation(len(ns)).tolist():
            k = co.lengths([j])[0]
            slot = st.slot(k)
            ks = co.feed([j], slot.x, slot.layout.bag_len_dev, slot.y)
            loss, prob = st.step(slot, ks, on_device=True)
            rows = oc.patch_drop_select(ns[j], keep_count(ns[j], keeps[j]), j, co.seed, epoch)
            o = orc.image_only_forward(bags[j][rows], ref)
            rloss = orc.bce_loss(o["prob"], labels[j:j + 1])
            opt.zero_grad()
            rloss.backward()
            torch.cuda.synchronize()
            assert abs(float(loss.item()) - float(rloss)) <= 2e-5, (epoch, j)
            assert float((tr.last["logits"].cpu() - o["logits"]).abs().max()) <= 1e-3 * 0.05, (epoch, j)
            assert torch.equal(prob.cpu().argmax(-1), o["prob"].argmax(-1))
            opt.step()
    assert st.replays >= 24 - 2 * len(st.slots)


def test_fusion_steps_fed_from_the_cohort_match_the_oracle_on_the_same_rows():
    from mil_amd.fusion_step import RaggedFusionStepper
    from mil_amd.model.utils import get_model
    from mil_amd.optim import FlatAdam
    args = SimpleNamespace(modality=["pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL",
                           num_classes=2, learnablePrompt=0, n_ctx=4, clinical_features=["a", "b"], clip_layers=1,
                           cache_text=0)
    torch.manual_seed(11)
    m = get_model(args).to(DEV).eval()
    opt = FlatAdam([q for q in m.parameters() if q.requires_grad], lr=0.0, weight_decay=0.0, counted=True)   # lr 0: weights stay put
    st = RaggedFusionStepper(m, opt, B=1)
    rng = np.random.default_rng(3)
    ns = [int(v) for v in rng.integers(2000, 5000, size=6)]
    keeps = [0.9, 0.8, 0.9, 0.8, 0.9, 0.8]
    co, bags, labels, ids = _cohort(ns, F=768, keeps=keeps, seed=5)
    with torch.no_grad():
        co.set_text(m.clinic_extractor(ids.to(DEV)))                      # --cache_text 1 as a device table
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    co.draw_epoch(1)
    worst = 0.0
    for j in [0, 1, 2, 3, 4, 5, 0, 3, 5, 1]:
        k = co.lengths([j])[0]
        slot = st.slot(k)
        ks = co.feed([j], slot.x, slot.bucket.len_dev, slot.y, text_dst=slot.text)
        loss, prob, z = st.step(slot, ks, on_device=True)
        rows = oc.patch_drop_select(ns[j], keep_count(ns[j], keeps[j]), j, co.seed, 1)
        with torch.no_grad():
            o = orc.fused_forward(bags[j][rows], ids[j], sd)
        torch.cuda.synchronize()
        worst = max(worst, float((z.cpu() - o["logits"]).abs().max()))
        assert torch.equal(prob.cpu().argmax(-1), o["prob"].argmax(-1)), j
    assert worst <= 1e-3 * 0.05, worst
    assert st.replays >= 4


def test_full_size_cohort_properties():
    """200 bags N ~ U[2000, 15592] x 768 (5.5 GB, generated on the device): every bag's draw is int(n * keep) distinct
    ascending rows of that bag, the gathered bucket equals index_select on them, two epochs differ."""
    rng = np.random.default_rng(0)
    ns = [int(v) for v in rng.integers(2000, 15593, size=200)]
    keeps = [0.9 if i % 2 else 0.8 for i in range(200)]
    g = torch.Generator(device=DEV).manual_seed(1)
    big = torch.randn((sum(ns), 768), device=DEV, generator=g)
    off = np.concatenate([[0], np.cumsum(ns)])
    co = DeviceCohort(lambda j: big[off[j]:off[j + 1]], syn.make_labels(1, 200), DEV, keep=keeps, seed=9, lengths=ns)
    co.draw_epoch(0)
    s0 = co.sel.clone()
    co.draw_epoch(1)
    torch.cuda.synchronize()
    s1 = co.sel.cpu().numpy()
    assert not np.array_equal(s0.cpu().numpy(), s1)
    for j in range(200):
        part = s1[co.sel_off[j]:co.sel_off[j + 1]]
        assert len(part) == int(ns[j] * keeps[j]) and (np.diff(part) > 0).all()
        assert part[0] >= off[j] and part[-1] < off[j + 1]
    x = torch.zeros((16384, 768), device=DEV)
    for j in (0, 57, 199):
        ks = co.feed([j], x)
        want = big.index_select(0, co.sel[co.sel_off[j]:co.sel_off[j + 1]].long())
        assert torch.equal(x[:ks[0]], want)


def test_cohort_edge_cases_empty_and_tiny_bags_and_rejected_arguments():
    """Edge cases the reference's loader meets: a bag whose keep count rounds to zero (int(1 * 0.9) = 0), an empty bag, bags
    of 1 - 5 rows; and the feed's argument checks (destination too small, wrong width, missing side table)."""
    ns = [1, 0, 2, 5, 64]
    keeps = [0.9, 0.9, 0.8, 0.8, 1.0]
    g = torch.Generator().manual_seed(2)
    bags = [torch.randn((n, 32), generator=g) for n in ns]
    co = DeviceCohort(bags, syn.make_labels(1, len(ns)), DEV, keep=keeps, seed=3)
    assert co.k_train == [0, 0, 1, 4, 64]
    co.draw_epoch(7)
    torch.cuda.synchronize()
    want = oc.select_epoch(np.asarray(co.row_off), co.k_train, co.seed, 7)
    assert np.array_equal(co.sel.cpu().numpy()[:co.sel_off[-1]], want)
    x = torch.full((128, 32), -1.0, device=DEV)
    ld = torch.full((5,), -1, device=DEV, dtype=torch.int32)
    ks = co.feed([0, 1, 2, 3, 4], x, ld)
    torch.cuda.synchronize()
    assert ks == [0, 0, 1, 4, 64] and ld.cpu().tolist() == ks
    flat = torch.cat(bags, 0)
    assert torch.equal(x[:69].cpu(), flat[want]) and bool((x[69:] == -1.0).all())
    with pytest.raises(ValueError):
        co.feed([4, 4, 4], torch.zeros((100, 32), device=DEV))           # 192 rows do not fit
    with pytest.raises(ValueError):
        co.feed([4], torch.zeros((128, 64), device=DEV))                 # wrong row width
    with pytest.raises(ValueError):
        co.feed([4], x, ids_dst=torch.zeros((1, 1, 77), device=DEV, dtype=torch.int64))      # the cohort holds no ids
    with pytest.raises(ValueError):
        DeviceCohort([torch.randn(4, 30)], syn.make_labels(1, 1), DEV)   # width not a multiple of 4 floats
