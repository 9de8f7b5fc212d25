"""GPU: the fused training step (trainer.ImageOnlyTrainer: gate -> pool -> fused tail -> backward -> Adam)
against the oracle + torch.optim.Adam on the CPU, over several steps."""
import pytest
import torch

from conftest import rel_err
from mil_amd import synthetic as syn
from mil_amd.bags import BagLayout
from mil_amd.trainer import ImageOnlyTrainer
from oracle import mil_oracle as orc

pytestmark = pytest.mark.gpu


def test_train_steps_match_oracle_adam():
    dev = torch.device("cuda")
    L, lengths = 512, [64, 200, 31, 129]
    p = syn.image_only_params(77, L=L)
    bags = [torch.randn((n, L), generator=torch.Generator().manual_seed(700 + i)) for i, n in enumerate(lengths)]
    y = syn.make_labels(70, len(lengths))
    lr = 1e-3       # larger than the reference's 1e-5 so three steps move the weights measurably
    tr = ImageOnlyTrainer(p, dev, lr=lr)
    lay = BagLayout.make(lengths, dev)
    x = torch.cat(bags, 0).to(dev)
    yd = y.to(dev)

    ref = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    opt = torch.optim.Adam(list(ref.values()), lr=lr, betas=(0.9, 0.999), weight_decay=1e-7)
    for step in range(3):
        loss_sum, prob = tr.train_step(x, lay, yd)
        outs = [orc.image_only_forward(b, ref) for b in bags]
        rprob = torch.cat([o["prob"] for o in outs], 0)
        rloss = orc.bce_loss(rprob, y)
        opt.zero_grad()
        rloss.backward()
        assert abs(float(loss_sum.item()) - float(rloss)) <= 1e-5, step
        assert torch.equal(prob.cpu().argmax(-1), rprob.argmax(-1))
        for k in ref:
            if float(ref[k].grad.norm()) > 1e-7:
                assert rel_err(tr.fp.g(k).cpu(), ref[k].grad) <= 2e-4, (step, k)
        noise_only = {k for k in ref if float(ref[k].grad.norm()) <= 1e-7}   # softmax bias: d/db == 0 exactly,
        opt.step()                                                          # Adam turns rounding noise into +-lr
        for k in ref:
            if k in noise_only:
                continue
            # Adam normalises by sqrt(v): elements whose gradient is ~eps amplify its relative error, so the
            # bound is 1 % of one lr-sized update, not the 1e-7 of the oracle's own Adam test
            assert float((tr.fp.p(k).cpu() - ref[k].detach()).abs().max()) <= 1e-2 * lr, (step, k)


def test_inference_path_has_no_labels():
    dev = torch.device("cuda")
    p = syn.image_only_params(78, L=512)
    tr = ImageOnlyTrainer(p, dev)
    x = torch.randn(300, 512, generator=torch.Generator().manual_seed(1))
    prob, z = tr.forward(x.to(dev), BagLayout.make([300], dev))
    o = orc.image_only_forward(x, p)
    assert float((z.cpu() - o["logits"]).abs().max()) <= 2e-5


def test_graph_replay_matches_eager():
    dev = torch.device("cuda")
    L, lengths = 512, [100, 40]
    p = syn.image_only_params(79, L=L)
    x = torch.randn(sum(lengths), L, generator=torch.Generator().manual_seed(2)).to(dev)
    y = syn.make_labels(71, 2).to(dev)
    lay = BagLayout.make(lengths, dev)
    a = ImageOnlyTrainer(p, dev, lr=1e-3)
    b = ImageOnlyTrainer(p, dev, lr=1e-3)
    b.capture(x, lay, y)
    for _ in range(3):
        la, _ = a.train_step(x, lay, y)
        lb, _ = b.replay_step()
        assert float((la - lb).abs()) == 0.0
    assert torch.equal(a.fp.flat, b.fp.flat)
