"""GPU: edge cases of the K1 path the reference's data can produce: the largest bag of the cohort (15 592 patches,
dataset.py:386-391), a bag without patches, the raw 768-d CTransPath width (aggregator_clip.py:36), more than two
classes, a one-bag batch."""
import pytest
import torch

from conftest import rel_err
from mil_amd import synthetic as syn
from mil_amd.bags import BagLayout
from mil_amd.trainer import ImageOnlyTrainer
from oracle import mil_oracle as orc

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _step(p, bags, y, L, **kw):
    tr = ImageOnlyTrainer(p, DEV, lr=1e-3, **kw)
    lengths = [b.shape[0] for b in bags]
    x = torch.cat(bags, 0).to(DEV) if sum(lengths) else torch.zeros((0, L), device=DEV)
    lay = BagLayout.make(lengths, DEV)
    prob, z = tr.forward(x, lay, y.to(DEV))
    tr.backward()
    torch.cuda.synchronize()
    return tr, prob, z


def test_largest_bag_of_the_cohort():
    L, N = 512, 15592
    p = syn.image_only_params(101, L=L)
    bags = [torch.randn((N, L), generator=torch.Generator().manual_seed(1)),
            torch.randn((37, L), generator=torch.Generator().manual_seed(2))]
    y = syn.make_labels(3, 2)
    tr, prob, z = _step(p, bags, y, L)
    loss, logits, rprob, grads = orc.batch_loss_and_grads(bags, y, p)
    assert float((z.cpu() - logits).abs().max()) <= 2e-5
    assert torch.equal(prob.cpu().argmax(-1), rprob.argmax(-1))
    for k in grads:
        if float(grads[k].norm()) > 1e-7:
            assert rel_err(tr.fp.g(k).cpu(), grads[k]) <= 2e-4, k


def test_raw_768_wide_features_and_three_classes():
    L = 768
    p = syn.image_only_params(102, L=L, C=3)
    bags = [torch.randn((n, L), generator=torch.Generator().manual_seed(10 + i)) for i, n in enumerate([130, 64, 5])]
    y = syn.make_labels(4, 3, C=3)
    tr, prob, z = _step(p, bags, y, L, loss="bce")
    loss, logits, rprob, grads = orc.batch_loss_and_grads(bags, y, p)
    assert float((z.cpu() - logits).abs().max()) <= 2e-5
    assert abs(float(tr.loss_sum.item()) - float(loss)) <= 1e-5
    for k in grads:
        if float(grads[k].norm()) > 1e-7:
            assert rel_err(tr.fp.g(k).cpu(), grads[k]) <= 2e-4, k


def test_bag_without_patches_is_harmless():
    """The reference would fail on an empty bag; here it pools to zero (logits = head bias), contributes a finite
    loss and no gate gradient, and leaves its neighbours untouched."""
    L = 512
    p = syn.image_only_params(103, L=L)
    full = [torch.randn((50, L), generator=torch.Generator().manual_seed(20)),
            torch.randn((70, L), generator=torch.Generator().manual_seed(21))]
    y3 = syn.make_labels(5, 3)
    tr, prob, z = _step(p, [full[0], torch.zeros((0, L)), full[1]], y3, L)
    assert torch.isfinite(z).all() and torch.isfinite(tr.fp.grad).all()
    assert float((z[1].cpu() - p["fc.1.bias"]).abs().max()) <= 1e-7
    for i, b in ((0, 0), (2, 1)):
        o = orc.image_only_forward(full[b], p)
        assert float((z[i].cpu() - o["logits"][0]).abs().max()) <= 2e-5


def test_single_bag_single_patch():
    L = 512
    p = syn.image_only_params(104, L=L)
    bags = [torch.randn((1, L), generator=torch.Generator().manual_seed(30))]
    y = syn.make_labels(6, 1)
    tr, prob, z = _step(p, bags, y, L)
    loss, logits, rprob, grads = orc.batch_loss_and_grads(bags, y, p)
    assert float((z.cpu() - logits).abs().max()) <= 2e-5
    # one patch: the softmax weight is 1, so no gradient reaches the gate parameters
    assert float(tr.fp.g("aggregator.attention_V.0.weight").abs().max()) <= 1e-7
    assert rel_err(tr.fp.g("fc.1.weight").cpu(), grads["fc.1.weight"]) <= 1e-5


def test_rows_beyond_a_whole_round_of_tiles_take_the_few_rows_path():
    """R = 256 * 128 + 40: the 40 rows run through mil_linear_small_fwd / k_gate_bwd_dx_tail instead of costing a
    second round of the grid.  Their scores, gates and dx must agree with the tiled kernels run on those rows alone."""
    from mil_amd import ops
    L, R, T = 512, 256 * 128 + 40, 40
    p = {k: v.to(DEV) for k, v in syn.image_only_params(105, L=L).items()}
    gp = [p["aggregator.attention_V.0.weight"], p["aggregator.attention_V.0.bias"], p["aggregator.attention_U.0.weight"],
          p["aggregator.attention_U.0.bias"], p["aggregator.attention_weights.weight"].reshape(-1),
          p["aggregator.attention_weights.bias"]]
    x = torch.randn((R, L), generator=torch.Generator().manual_seed(9)).to(DEV)
    scores, gates = ops.gate_scores_fwd(x, *gp, save_gates=True)
    s_ref, g_ref = ops.gate_scores_fwd(x[R - T:].contiguous(), *gp, save_gates=True)
    assert float((scores[R - T:] - s_ref).abs().max()) <= 2e-6
    assert float((gates[R - T:] - g_ref).abs().max()) <= 2e-6
    s_head, _ = ops.gate_scores_fwd(x[:128].contiguous(), *gp, save_gates=True)
    assert float((scores[:128] - s_head).abs().max()) <= 1e-6
    ds = torch.randn((R,), generator=torch.Generator().manual_seed(10)).to(DEV)
    dx = torch.zeros((R, L), device=DEV)
    ops.gate_bwd_input(gates, ds, gp[4], gp[0], gp[2], dx)
    dx_ref = torch.zeros((T, L), device=DEV)
    ops.gate_bwd_input(g_ref, ds[R - T:].contiguous(), gp[4], gp[0], gp[2], dx_ref)
    assert rel_err(dx[R - T:].cpu(), dx_ref.cpu()) <= 2e-6
    assert float(dx[:R - T].abs().sum()) > 0


def test_a_last_round_of_a_few_tiles_goes_through_the_32_row_kernel():
    """R = 256 * 128 + 320 (32 bags x (1024 patches + 10 text tokens)): the 320 rows beyond the whole round run on
    k_gate_fwd_r32 instead of as a second, nearly empty round of 128-row tiles.  Same scores / gates as the tiled kernel on
    those rows alone, in eval mode and with keep bits."""
    from mil_amd import ops
    L, T = 512, 320
    R = 256 * 128 + T
    p = {k: v.to(DEV) for k, v in syn.image_only_params(106, L=L).items()}
    gp = [p["aggregator.attention_V.0.weight"], p["aggregator.attention_V.0.bias"], p["aggregator.attention_U.0.weight"],
          p["aggregator.attention_U.0.bias"], p["aggregator.attention_weights.weight"].reshape(-1),
          p["aggregator.attention_weights.bias"]]
    x = torch.randn((R, L), generator=torch.Generator().manual_seed(11)).to(DEV)
    for bits in (None, torch.randint(0, 2 ** 31 - 1, (R, L // 32), generator=torch.Generator().manual_seed(12),
                                     dtype=torch.int32).to(DEV)):
        kw = {} if bits is None else dict(xbits=bits, xscale=2.0)
        scores, gates = ops.gate_scores_fwd(x, *gp, save_gates=True, **kw)
        kt = {} if bits is None else dict(xbits=bits[R - T:].contiguous(), xscale=2.0)
        s_ref, g_ref = ops.gate_scores_fwd(x[R - T:].contiguous(), *gp, save_gates=True, **kt)
        assert float((scores[R - T:] - s_ref).abs().max()) <= 2e-6
        assert float((gates[R - T:] - g_ref).abs().max()) <= 2e-6
        kh = {} if bits is None else dict(xbits=bits[:256].contiguous(), xscale=2.0)
        s_head, _ = ops.gate_scores_fwd(x[:256].contiguous(), *gp, save_gates=True, **kh)
        assert float((scores[:256] - s_head).abs().max()) <= 2e-6
        s_nog, _ = ops.gate_scores_fwd(x, *gp, save_gates=False, **kw)          # without saved gates: same split, same scores
        assert float((s_nog - scores).abs().max()) <= 2e-6


def test_row_tiles_per_workgroup_of_the_32_row_kernel_agree():
    """k_gate_fwd_r32<., RT>: 20 000 rows take three row tiles per workgroup, their halves two, a 4 096-row slice one;
    scores and gates of the same rows must agree whatever the tiling (eval mode and with keep bits)."""
    from mil_amd import ops
    L, R = 512, 20000
    p = {k: v.to(DEV) for k, v in syn.image_only_params(107, L=L).items()}
    gp = [p["aggregator.attention_V.0.weight"], p["aggregator.attention_V.0.bias"], p["aggregator.attention_U.0.weight"],
          p["aggregator.attention_U.0.bias"], p["aggregator.attention_weights.weight"].reshape(-1),
          p["aggregator.attention_weights.bias"]]
    x = torch.randn((R, L), generator=torch.Generator().manual_seed(13)).to(DEV)
    bits = torch.randint(0, 2 ** 31 - 1, (R, L // 32), generator=torch.Generator().manual_seed(14), dtype=torch.int32).to(DEV)
    for kw_of in (lambda a, b: {}, lambda a, b: dict(xbits=bits[a:b].contiguous(), xscale=2.0)):
        s3, g3 = ops.gate_scores_fwd(x, *gp, save_gates=True, **kw_of(0, R))
        for a, b in ((0, 10000), (10000, R), (3000, 7096)):
            s_, g_ = ops.gate_scores_fwd(x[a:b].contiguous(), *gp, save_gates=True, **kw_of(a, b))
            assert float((s3[a:b] - s_).abs().max()) <= 2e-6
            assert float((g3[a:b] - g_).abs().max()) <= 2e-6
    assert bool(torch.isfinite(s3).all())


def test_more_than_two_classes_use_cross_entropy_on_the_sigmoid_outputs():
    """num_classes > 2: the reference's criterion is CrossEntropyLoss applied to the module's sigmoid outputs with the
    float one-hot labels as class probabilities (train_ddp.py:95-96,323-324)."""
    L = 512
    p = syn.image_only_params(104, L=L, C=3)
    bags = [torch.randn((n, L), generator=torch.Generator().manual_seed(20 + i)) for i, n in enumerate([90, 33, 64, 7])]
    y = syn.make_labels(5, 4, C=3)
    tr, prob, z = _step(p, bags, y, L)                      # default criterion for C = 3
    assert tr.loss == "ce"
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    outs = [orc.image_only_forward(b, leaves) for b in bags]
    rprob = torch.cat([o["prob"] for o in outs], 0)
    rloss = torch.nn.CrossEntropyLoss()(rprob, y)
    rloss.backward()
    assert abs(float(tr.loss_sum.item()) - float(rloss)) <= 1e-5
    assert torch.equal(prob.cpu().argmax(-1), rprob.argmax(-1))
    for k in leaves:
        if leaves[k].grad is not None and float(leaves[k].grad.norm()) > 1e-7:
            assert rel_err(tr.fp.g(k).cpu(), leaves[k].grad) <= 2e-4, k
