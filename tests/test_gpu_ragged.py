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
