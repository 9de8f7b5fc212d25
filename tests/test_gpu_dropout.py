"""GPU: train-mode dropout of the hot path (ABMIL.py:49 Dropout(.5) on the bag, aggregator.py:129 Dropout(.25) in front
of the head) done in-kernel through Philox keep-bit tensors.  The masks are an explicit output of the step, so the
oracle is evaluated on EXACTLY the masks the kernels used: same bars as the eval-mode parity tests."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from mil_amd import ops, synthetic as syn
from mil_amd.bags import BagLayout
from mil_amd.trainer import ImageOnlyTrainer
from oracle import mil_oracle as orc
from oracle import philox as P

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _u32(t):
    return t.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("p_drop", [0.5, 0.25, 0.1])
def test_keep_bits_equal_the_numpy_philox(p_drop):
    rows, cols = 77, 768
    bits = ops.dropout_keep_bits(rows, cols, p_drop, seed=(5 << 32) | 1234, offset=(2 << 32) | 9, device=DEV)
    assert np.array_equal(_u32(bits), P.keep_bits(rows, cols, p_drop, (5 << 32) | 1234, (2 << 32) | 9))
    ctr = torch.tensor([4], device=DEV, dtype=torch.int32)
    b2 = ops.dropout_keep_bits(rows, cols, p_drop, 1234, 3, DEV, offset_dev=ctr)
    assert np.array_equal(_u32(b2), P.keep_bits(rows, cols, p_drop, 1234, 7))
    ops.counter_add(ctr, 2)
    assert int(ctr.item()) == 6


def _masked_oracle(bags, y, p, xbits, mbits, L):
    kx = torch.from_numpy(P.unpack_bits(_u32(xbits), L))
    km = torch.from_numpy(P.unpack_bits(_u32(mbits), L))
    off = np.cumsum([0] + [b.shape[0] for b in bags])
    names = list(p.keys())
    leaves = {k: p[k].clone().requires_grad_(True) for k in names}
    outs = [orc.image_only_forward(b, leaves, keep_x=kx[off[i]:off[i + 1]], keep_m=km[i:i + 1]) for i, b in enumerate(bags)]
    prob = torch.cat([o["prob"] for o in outs], 0)
    logits = torch.cat([o["logits"] for o in outs], 0)
    loss = orc.bce_loss(prob, y)
    g = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    return loss.detach(), logits.detach(), prob.detach(), {k: (gi if gi is not None else torch.zeros_like(p[k])) for k, gi in zip(names, g)}


@pytest.mark.parametrize("lengths,L", [([300, 77, 1, 129], 512), ([130, 64], 768), ([20000, 12808], 512),
                                       ([16384, 16000, 384], 512), ([20000, 4576], 1024)])
def test_train_mode_step_matches_oracle_on_the_same_masks(lengths, L):
    """[20000, 12808] = 256 x 128 + 40 rows: the 128-row MFMA kernel with the mask on its A fragments plus the 32-row
    kernel on the 40 rows beyond whole rounds (keep bits from the stand-alone generator); the short cases run the 32-row
    kernel only; 32 768 and 24 576 rows = whole rounds of 128-row tiles: the forward kernel draws the keep bits itself
    (mil_gate_scores_fwd_draw) - the same words, checked against the numpy Philox."""
    p = syn.image_only_params(31, L=L)
    bags = [torch.randn((n, L), generator=torch.Generator().manual_seed(400 + i)) for i, n in enumerate(lengths)]
    y = syn.make_labels(32, len(lengths))
    tr = ImageOnlyTrainer(p, DEV, lr=1e-3, train_mode=True, seed=99)
    x = torch.cat(bags, 0).to(DEV)
    lay = BagLayout.make(lengths, DEV)
    prob, z = tr.forward(x, lay, y.to(DEV))
    tr.backward()
    torch.cuda.synchronize()
    xbits, mbits = tr.last["xbits"], tr.last["mbits"]
    frac = float(torch.from_numpy(P.unpack_bits(_u32(xbits), L)).mean())
    assert 0.48 < frac < 0.52
    assert np.array_equal(_u32(xbits), P.keep_bits(sum(lengths), L, 0.5, 99, 1))          # first pass: stream position 1
    assert np.array_equal(_u32(mbits), P.keep_bits(len(lengths), L, 0.25, 99 ^ 0x9E3779B97F4A7C15, 1))
    loss, logits, rprob, grads = _masked_oracle(bags, y, p, xbits, mbits, L)
    assert float((z.cpu() - logits).abs().max()) <= 2e-5                                   # bar: 1e-3
    assert torch.equal(prob.cpu().argmax(-1), rprob.argmax(-1))
    assert abs(float(tr.loss_sum.item()) - float(loss)) <= 1e-5
    for k in grads:
        if float(grads[k].norm()) > 1e-7:
            assert rel_err(tr.fp.g(k).cpu(), grads[k]) <= 2e-4, (k, rel_err(tr.fp.g(k).cpu(), grads[k]))
    # a second pass draws another mask; eval-mode inference ignores dropout
    first = _u32(xbits).copy()
    tr.forward(x, lay, y.to(DEV))
    assert not np.array_equal(_u32(tr.last["xbits"]), first)
    pe, ze = tr.forward(x, lay, None)
    o = torch.cat([orc.image_only_forward(b, p)["logits"] for b in bags], 0)
    assert float((ze.cpu() - o).abs().max()) <= 2e-5


def test_masks_change_every_pass_and_every_micro_batch():
    L, lengths = 512, [64, 64]
    p = syn.image_only_params(33, L=L)
    x = torch.randn(128, L, device=DEV)
    y = syn.make_labels(1, 2).to(DEV)
    lay = BagLayout.make(lengths, DEV)
    tr = ImageOnlyTrainer(p, DEV, train_mode=True, accum=2)
    seen = []
    for _ in range(4):
        tr.train_step(x, lay, y)
        seen.append(_u32(tr.last["xbits"]).copy())
    for i in range(4):
        for j in range(i):
            assert not np.array_equal(seen[i], seen[j])


def test_autograd_route_with_keep_bits_incl_input_gradient():
    """ops.gated_attention_pool(..., xbits): M and every gradient, dx included (mask applied by the last writer of dx)."""
    L, lengths = 512, [200, 57]
    p = syn.image_only_params(35, L=L)
    d = {k: v.to(DEV).requires_grad_(True) for k, v in p.items()}
    bags = [torch.randn((n, L), generator=torch.Generator().manual_seed(60 + i)) for i, n in enumerate(lengths)]
    x = torch.cat(bags, 0).to(DEV).requires_grad_(True)
    lay = BagLayout.make(lengths, DEV)
    xbits = ops.dropout_keep_bits(sum(lengths), L, 0.5, 11, 0, DEV)
    M, _ = ops.gated_attention_pool(x, d["aggregator.attention_V.0.weight"], d["aggregator.attention_V.0.bias"],
                                    d["aggregator.attention_U.0.weight"], d["aggregator.attention_U.0.bias"],
                                    d["aggregator.attention_weights.weight"], d["aggregator.attention_weights.bias"], lay, xbits)
    tgt = torch.randn(M.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
    (M * tgt).sum().backward()
    keep = torch.from_numpy(P.unpack_bits(_u32(xbits), L))
    xr = torch.cat(bags, 0).requires_grad_(True)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    off = [0, lengths[0], sum(lengths)]
    Ms = [orc.abmil_forward(xr[off[i]:off[i + 1]], leaves, keep=keep[off[i]:off[i + 1]])[0] for i in range(2)]
    Mr = torch.cat(Ms, 0)
    (Mr * tgt.cpu()).sum().backward()
    assert float((M.detach().cpu() - Mr.detach()).abs().max()) <= 2e-5
    assert rel_err(x.grad.cpu(), xr.grad) <= 2e-4
    assert float((x.grad.cpu()[keep == 0]).abs().max()) == 0.0
    for k in ("aggregator.attention_V.0.weight", "aggregator.attention_U.0.weight", "aggregator.attention_weights.weight",
              "aggregator.attention_V.0.bias", "aggregator.attention_U.0.bias"):
        assert rel_err(d[k].grad.cpu(), leaves[k].grad) <= 2e-4, k


@pytest.mark.parametrize("grad_mfma", [False, True])
@pytest.mark.parametrize("lengths,L", [([300, 77, 129], 512), ([1000, 24], 1024)])
def test_bf16_train_mode_step_matches_oracle_on_the_same_masks(lengths, L, grad_mfma):
    """bf16 storage (BASELINE config 5's arithmetic) in model.train() mode: the bf16 kernels read x through the same keep-bit
    tensors - forward A fragments, pool pass, weight-gradient staging.  Oracle: fp32 on the bf16-rounded x and gate weights
    with exactly the masks the step drew."""
    p = syn.image_only_params(41, L=L)
    bags = [torch.randn((n, L), generator=torch.Generator().manual_seed(500 + i)) for i, n in enumerate(lengths)]
    y = syn.make_labels(42, len(lengths))
    tr = ImageOnlyTrainer(p, DEV, lr=1e-3, train_mode=True, seed=7, bf16_grad_mfma=grad_mfma)
    x16 = torch.cat(bags, 0).to(DEV).to(torch.bfloat16)
    lay = BagLayout.make(lengths, DEV)
    prob, z = tr.forward(x16, lay, y.to(DEV))
    tr.backward()
    torch.cuda.synchronize()
    xbits, mbits = tr.last["xbits"], tr.last["mbits"]
    assert np.array_equal(_u32(xbits), P.keep_bits(sum(lengths), L, 0.5, 7, 1))
    pr = dict(p)
    for k in ("aggregator.attention_V.0.weight", "aggregator.attention_U.0.weight"):
        pr[k] = p[k].to(torch.bfloat16).float()
    rb = [b.to(torch.bfloat16).float() for b in bags]
    loss, logits, rprob, grads = _masked_oracle(rb, y, pr, xbits, mbits, L)
    assert float((z.cpu() - logits).abs().max()) <= 5e-5
    assert torch.equal(prob.cpu().argmax(-1), rprob.argmax(-1))
    assert abs(float(tr.loss_sum.item()) - float(loss)) <= 1e-5
    for k in grads:
        if float(grads[k].norm()) > 1e-7:
            gate_param = k.startswith("aggregator.attention")
            tol = 1.2e-2 if (grad_mfma and L % 256 == 0 and gate_param) else \
                (6e-3 if (grad_mfma and k.endswith(("attention_V.0.weight", "attention_U.0.weight"))) else 5e-4)
            assert rel_err(tr.fp.g(k).cpu(), grads[k]) <= tol, (k, rel_err(tr.fp.g(k).cpu(), grads[k]))
    # three training steps run and keep every parameter finite
    for _ in range(3):
        tr.train_step(x16, lay, y.to(DEV))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(tr.fp.flat).all())


def test_image_only_module_in_train_mode_draws_philox_masks_for_patches_and_head():
    """model/aggregator_clip.py in `model.train()`: BOTH dropouts come from the module's Philox stream (the head
    used torch's generator while the fused route drew keep words), the masks are outputs, and logits + gradients equal the
    oracle on exactly those masks."""
    from types import SimpleNamespace
    from mil_amd.model.utils_clip import get_model
    L, lengths = 512, [128, 128, 128]
    torch.manual_seed(3)
    m = get_model(SimpleNamespace(modality=["pathology"], model_pathology="ABMIL", num_classes=2, patch_dim=L)).to(DEV).train()
    bags = [torch.randn((n, L), generator=torch.Generator().manual_seed(70 + i)) for i, n in enumerate(lengths)]
    pad = torch.stack(bags, 0)
    y = syn.make_labels(5, 3)
    _, prob = m([pad.to(DEV)])
    loss = torch.nn.BCELoss()(prob, y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    ab = m.extractor_pathology
    p = {k.replace("extractor_pathology.", "aggregator."): v.detach().cpu() for k, v in m.state_dict().items()}
    rl, rz, rp, rg = _masked_oracle(bags, y, p, ab.last_xbits, m.last_mbits, L)
    # the head's words are the fused tail's: key = module seed ^ golden ratio, stream position = the pass counter
    want = P.keep_bits(3, L, 0.25, (ab._drop_seed ^ 0x9E3779B97F4A7C15) & (2 ** 64 - 1), int(ab._drop_ctr.item()))
    assert np.array_equal(_u32(m.last_mbits), want)
    assert float((m.last_logits.detach().cpu() - rz).abs().max()) <= 2e-5
    assert abs(float(loss.item()) - float(rl)) <= 2e-6
    for k, v in m.named_parameters():
        ko = k.replace("extractor_pathology.", "aggregator.")
        if ko.endswith("attention_weights.bias"):
            continue
        assert rel_err(v.grad.cpu(), rg[ko]) <= 2e-4, k
    b1 = m.last_mbits.clone()
    m([pad.to(DEV)])
    assert not torch.equal(b1, m.last_mbits)                       # a fresh head mask every pass
