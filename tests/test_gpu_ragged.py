"""GPU: the authors' regime - ONE ragged bag per step whose length changes every step (run_train.sh:81; the train-time
patch drop of dataset.py:374-381).  Lengths live on the device, tile maps are rebuilt there, and a capacity bucket's step
is one captured hipGraph: a stream of 50 bags with 2 000 .. 15 592 patches must replay at most 8 graphs and match the
oracle bag by bag."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from mil_amd import synthetic as syn
from mil_amd.bags import BagLayout, DeviceBagLayout, bucket_rows
from mil_amd.trainer import ImageOnlyTrainer, RaggedImageOnlyStepper
from oracle import mil_oracle as orc

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def test_device_built_tile_map_equals_the_host_one():
    from mil_amd import _lib, ops
    lengths = [300, 1, 0, 77, 4096, 33]
    lay = DeviceBagLayout(bucket_rows(sum(lengths)), len(lengths), DEV).set_lengths(lengths)
    rc = _lib.lib().mil_build_tile_map(ops._p(lay.bag_len_dev), lay.B, ops._p(lay.tile_map), ops._p(lay.bag_tile_off),
                                       ops._p(lay.rows_dev), lay.T, ops._stream())
    assert rc == 0
    ref = BagLayout.make(lengths, DEV)
    assert int(lay.rows_dev.item()) == sum(lengths)
    assert torch.equal(lay.bag_tile_off.cpu(), ref.bag_tile_off.cpu())
    assert torch.equal(lay.tile_map[:ref.T].cpu(), ref.tile_map.cpu())
    assert int(lay.tile_map[ref.T:].abs().sum()) == 0                   # padding tiles: nrows == 0


def test_stream_of_ragged_bags_replays_few_graphs_and_matches_the_oracle():
    L, lr = 512, 1e-3
    p = syn.image_only_params(41, L=L)
    tr = ImageOnlyTrainer(p, DEV, lr=lr, counted=True)                  # eval-mode arithmetic: comparable with the oracle
    st = RaggedImageOnlyStepper(tr, B=1)
    rng = np.random.default_rng(5)
    ref = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    opt = torch.optim.Adam(list(ref.values()), lr=lr, betas=(0.9, 0.999), weight_decay=1e-7)
    checked = 0
    for step in range(50):
        n = int(rng.integers(2000, 15593))
        x = torch.randn((n, L), generator=torch.Generator().manual_seed(1000 + step))
        y = syn.make_labels(2000 + step, 1)
        slot = st.slot(n)
        slot.x[:n].copy_(x.to(DEV))
        slot.y.copy_(y.to(DEV))
        loss, prob = st.step(slot, [n])
        # the fused update overwrites the pre-step weights: the oracle + torch.optim.Adam run in lock-step instead
        o = orc.image_only_forward(x, ref)
        rloss = orc.bce_loss(o["prob"], y)
        opt.zero_grad()
        rloss.backward()
        torch.cuda.synchronize()
        assert abs(float(loss.item()) - float(rloss)) <= 2e-5, (step, n)
        assert float((tr.last["logits"].cpu() - o["logits"]).abs().max()) <= 5e-5, (step, n)
        assert torch.equal(prob.cpu().argmax(-1), o["prob"].argmax(-1))
        opt.step()
        checked += 1
    assert checked == 50
    assert len(st.slots) <= 8 and sum(s.graph is not None for s in st.slots.values()) <= 8
    assert st.replays >= 50 - 2 * len(st.slots) and st.replays + st.eager_steps == 50
    for k in ref:
        if k.endswith("attention_weights.bias"):
            continue                                                      # softmax bias: zero gradient, Adam noise
        assert float((tr.fp.p(k).cpu() - ref[k].detach()).abs().max()) <= 50 * 1e-2 * lr, k


def test_bucketed_step_in_train_mode_matches_oracle_on_its_masks():
    """Device-side lengths + dropout: the masks cover the bucket, the oracle uses the true rows of them."""
    from oracle import philox as P
    L, n = 512, 2500
    p = syn.image_only_params(43, L=L)
    tr = ImageOnlyTrainer(p, DEV, lr=1e-3, train_mode=True, seed=7, counted=True)
    st = RaggedImageOnlyStepper(tr, B=1, use_graph=False)
    x = torch.randn((n, L), generator=torch.Generator().manual_seed(3))
    y = syn.make_labels(4, 1)
    slot = st.slot(n)
    slot.x[:n].copy_(x.to(DEV))
    slot.y.copy_(y.to(DEV))
    slot.layout.set_lengths([n])
    prob, z = tr.forward(slot.x, slot.layout, slot.y)
    tr.backward()
    torch.cuda.synchronize()
    kx = torch.from_numpy(P.unpack_bits(tr.last["xbits"].cpu().numpy().view(np.uint32), L))[:n]
    km = torch.from_numpy(P.unpack_bits(tr.last["mbits"].cpu().numpy().view(np.uint32), L))
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    o = orc.image_only_forward(x, leaves, keep_x=kx, keep_m=km)
    loss = orc.bce_loss(o["prob"], y)
    loss.backward()
    assert float((z.cpu() - o["logits"]).abs().max()) <= 2e-5
    for k in leaves:
        g = leaves[k].grad
        if g is not None and float(g.norm()) > 1e-7:
            assert rel_err(tr.fp.g(k).cpu(), g) <= 2e-4, k


def _lockstep(tr, st, lengths, lr_of_step, L, seed0=3000):
    """Run the stepper over `lengths` with torch.optim.Adam + the oracle in lock-step; returns the reference leaves."""
    ref = {k: v.clone().requires_grad_(True) for k, v in syn.image_only_params(41, L=L).items()}
    opt = torch.optim.Adam(list(ref.values()), lr=lr_of_step(0), betas=(0.9, 0.999), weight_decay=1e-7)
    lr_sum = 0.0
    for step, n in enumerate(lengths):
        lr = lr_of_step(step)
        tr.lr = lr                                                         # what train_ddp.py does once per epoch
        for gph in opt.param_groups:
            gph["lr"] = lr
        x = torch.randn((n, L), generator=torch.Generator().manual_seed(seed0 + step))
        y = syn.make_labels(seed0 + 500 + step, 1)
        slot = st.slot(n)
        slot.x[:n].copy_(x.to(DEV))
        slot.y.copy_(y.to(DEV))
        loss, prob = st.step(slot, [n])
        o = orc.image_only_forward(x, ref)
        rloss = orc.bce_loss(o["prob"], y)
        opt.zero_grad()
        rloss.backward()
        torch.cuda.synchronize()
        assert abs(float(loss.item()) - float(rloss)) <= 2e-5, (step, n)
        assert float((tr.last["logits"].cpu() - o["logits"]).abs().max()) <= 5e-5, (step, n)
        opt.step()
        lr_sum += lr
        for k in ref:
            if k.endswith("attention_weights.bias"):
                continue
            # a learning rate frozen at capture time would be off by ~lr of the first epoch per step
            assert float((tr.fp.p(k).cpu() - ref[k].detach()).abs().max()) <= 0.05 * lr_sum, (step, k)
    return ref


def test_learning_rate_schedule_reaches_the_replayed_graph():
    """Adam runs inside the captured graph at world size 1; the scheduled rate (utils.py:232-241) must
    still reach it - it is read from a device word, not from a by-value kernel argument frozen at capture."""
    L = 512
    tr = ImageOnlyTrainer(syn.image_only_params(41, L=L), DEV, lr=1e-3, counted=True)
    st = RaggedImageOnlyStepper(tr, B=1)
    sched = lambda s: 1e-3 if s < 3 else (1e-5 if s < 6 else 3e-4)         # noqa: E731
    _lockstep(tr, st, [2100, 2300, 2200, 2500, 2050, 2400, 2150, 2250], sched, L)
    assert st.replays >= 6 and len(st.slots) == 1


def test_two_buckets_with_the_optimizer_outside_the_graph():
    """with an all-reduce between gradient and update (world > 1 / forced collectives) Adam stays outside
    the graph; a replay of bucket A must not leave A's descriptor behind for bucket B's capture."""
    L = 512
    tr = ImageOnlyTrainer(syn.image_only_params(41, L=L), DEV, lr=1e-3, counted=True)
    tr.force_collectives = True                                            # no process group here: the collective is a no-op
    st = RaggedImageOnlyStepper(tr, B=1)
    # bucket 3072 eager, bucket 2048 eager, 2048 capture + replay, 3072 capture (the stale-struct hazard), then both replay
    lengths = [2900, 1900, 2000, 3000, 1800, 2800, 2040, 3050]
    _lockstep(tr, st, lengths, lambda s: 1e-3, L)
    assert len(st.slots) == 2 and all(s.graph is not None and not s.graph["adam"] for s in st.slots.values())


def test_capture_refuses_gradient_accumulation():
    from mil_amd._lib import MilHipError
    L = 512
    tr = ImageOnlyTrainer(syn.image_only_params(41, L=L), DEV, counted=True, accum=2)
    x = torch.zeros((64, L), device=DEV)
    with pytest.raises(MilHipError):
        tr.capture(x, BagLayout.make([64], DEV), syn.make_labels(1, 1).to(DEV))
