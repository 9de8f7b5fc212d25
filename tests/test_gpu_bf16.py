"""GPU: the bf16-storage variant of K1 (BASELINE config 5).  Exactness reference: the fp32 oracle evaluated on the
SAME bf16-rounded inputs (x and gate weights rounded to bf16, everything else fp32) - against that the kernels must
match to fp32 accuracy.  Against the unrounded fp32 oracle the deviation is reported and bounded loosely."""
import pytest
import torch

from conftest import rel_err
from mil_amd import ops, synthetic as syn
from mil_amd.bags import BagLayout
from mil_amd.trainer import ImageOnlyTrainer
from oracle import mil_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _round(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("grad_mfma", [False, True])
@pytest.mark.parametrize("lengths,L", [([300, 77], 512), ([1000], 1024), ([33, 1, 129], 1024)])
def test_bf16_step_matches_oracle_on_rounded_inputs(lengths, L, grad_mfma):
    p = syn.image_only_params(55, L=L)
    bags = [torch.randn((n, L), generator=torch.Generator().manual_seed(800 + i)) for i, n in enumerate(lengths)]
    y = syn.make_labels(56, len(lengths))
    tr = ImageOnlyTrainer(p, torch.device(DEV), lr=1e-3, bf16_grad_mfma=grad_mfma)
    x16 = torch.cat(bags, 0).to(DEV).to(torch.bfloat16)
    lay = BagLayout.make(lengths, torch.device(DEV))
    prob, z = tr.forward(x16, lay, y.to(DEV))
    tr.backward()
    # oracle on rounded x; the gate weights are rounded only inside the gate (the pool and dW use fp32 masters... the
    # gradient flows to the fp32 master through the identity, as in mixed-precision training)
    pr = dict(p)
    pr["aggregator.attention_V.0.weight"] = _round(p["aggregator.attention_V.0.weight"])
    pr["aggregator.attention_U.0.weight"] = _round(p["aggregator.attention_U.0.weight"])
    rb = [_round(b) for b in bags]
    loss, logits, rprob, grads = orc.batch_loss_and_grads(rb, y, pr)
    assert float((z.cpu() - logits).abs().max()) <= 5e-5
    assert torch.equal(prob.cpu().argmax(-1), rprob.argmax(-1))
    assert abs(float(tr.loss_sum.item()) - float(loss)) <= 1e-5
    for k in grads:
        if float(grads[k].norm()) > 1e-7:
            # fp32-MFMA gradient: exact on the rounded x.  bf16-MFMA gradient: dPre is rounded to bf16 too
            # (relative 2^-9 per element), only the two weight matrices see it - biases / w / head stay fp32.
            # With the bf16-MFMA gradient the saved gates V, U are bf16 too (half the bytes of the step's largest tensor), so the
            # gate biases and w see 2^-9 relative rounding per factor as well; the head and the fp32-MFMA variant stay fp32.
            gate_param = k.startswith("aggregator.attention")
            tol = 1.2e-2 if (grad_mfma and L % 256 == 0 and gate_param) else \
                (6e-3 if (grad_mfma and k.endswith(("attention_V.0.weight", "attention_U.0.weight"))) else 5e-4)
            assert rel_err(tr.fp.g(k).cpu(), grads[k]) <= tol, (k, rel_err(tr.fp.g(k).cpu(), grads[k]))
    # deviation from the UNROUNDED fp32 oracle (reported in DESIGN.md; loose bound here)
    _, logits32, _, _ = orc.batch_loss_and_grads(bags, y, p)
    dev = float((z.cpu() - logits32).abs().max())
    print(f"bf16 vs fp32 oracle: max |dlogit| = {dev:.2e}")
    assert dev <= 5e-3


def test_cast_bf16_round_to_nearest_even():
    x = torch.tensor([1.0, 1.00390625, 1.005859375, -2.5, 3.0e38, 1e-40, float("inf")] + [0.1 * i for i in range(25)])
    got = ops.cast_bf16(x.to(DEV)).cpu()
    assert torch.equal(got, x.to(torch.bfloat16))


def test_deep_pipeline_kernel_matches_the_128_row_kernel():
    """R >= 65 536 switches mil_gate_scores_fwd_bf16 to the 256-row, two-ring LDS-DMA kernel: same k order, so its
    scores and gates must equal the 128-row kernel's (run on slices) bit for bit, including a ragged last tile."""
    from mil_amd import ops, synthetic as syn
    L, R = 1024, 65536 + 72
    p = {k: v.to(DEV) for k, v in syn.image_only_params(11, L=L).items()}
    x16 = ops.cast_bf16(torch.randn((R, L), generator=torch.Generator().manual_seed(3)).to(DEV))
    args = (ops.cast_bf16(p["aggregator.attention_V.0.weight"]), p["aggregator.attention_V.0.bias"],
            ops.cast_bf16(p["aggregator.attention_U.0.weight"]), p["aggregator.attention_U.0.bias"],
            p["aggregator.attention_weights.weight"].reshape(-1), p["aggregator.attention_weights.bias"])
    s_all, g_all = ops.gate_scores_fwd_bf16(x16, *args, save_gates=True)
    for lo, hi in ((0, 4096), (30000, 34000), (R - 1000, R)):
        s_ref, g_ref = ops.gate_scores_fwd_bf16(x16[lo:hi].contiguous(), *args, save_gates=True)
        assert torch.equal(s_all[lo:hi], s_ref)
        assert torch.equal(g_all[lo:hi], g_ref)


def test_deep_pipeline_kernel_with_keep_bits_matches_the_128_row_kernel():
    """Train mode at config-5 size: the deep kernel takes the dropout keep words through its own LDS-DMA stream (a three-group
    ring); scores and gates must equal the 128-row kernel's (run on slices with the matching slices of the bit tensor), bit
    for bit, including a ragged last tile."""
    from mil_amd import ops, synthetic as syn
    L, R = 1024, 65536 + 72
    p = {k: v.to(DEV) for k, v in syn.image_only_params(11, L=L).items()}
    x16 = ops.cast_bf16(torch.randn((R, L), generator=torch.Generator().manual_seed(3)).to(DEV))
    bits = ops.dropout_keep_bits(R, L, 0.5, seed=77, offset=3, device=torch.device(DEV))
    args = (ops.cast_bf16(p["aggregator.attention_V.0.weight"]), p["aggregator.attention_V.0.bias"],
            ops.cast_bf16(p["aggregator.attention_U.0.weight"]), p["aggregator.attention_U.0.bias"],
            p["aggregator.attention_weights.weight"].reshape(-1), p["aggregator.attention_weights.bias"])
    s_all, g_all = ops.gate_scores_fwd_bf16(x16, *args, save_gates=True, xbits=bits, xscale=2.0)
    s_eval, _ = ops.gate_scores_fwd_bf16(x16, *args, save_gates=False)
    assert not torch.equal(s_all, s_eval)                         # the mask does something
    for lo, hi in ((0, 4096), (30000, 34000), (R - 1000, R)):
        s_ref, g_ref = ops.gate_scores_fwd_bf16(x16[lo:hi].contiguous(), *args, save_gates=True,
                                                xbits=bits[lo:hi].contiguous(), xscale=2.0)
        assert torch.equal(s_all[lo:hi], s_ref)
        assert torch.equal(g_all[lo:hi], g_ref)
    s16, g16 = ops.gate_scores_fwd_bf16(x16, *args, save_gates=True, gates_bf16=True, xbits=bits, xscale=2.0)
    assert torch.equal(s16, s_all)
    assert torch.equal(g16, g_all.to(torch.bfloat16))


def test_fused_bf16_tail_equals_the_separate_launches_bit_for_bit():
    """World size 1: the fold launch of the bf16 weight gradient forms the head gradients, applies Adam and rewrites the bf16
    weight shadows (csrc/step.hip `tail16`).  Three steps that way must leave exactly the bits of forward / backward /
    stand-alone Adam + cast launches."""
    L, lengths = 1024, [700, 300, 1048]
    p = syn.image_only_params(77, L=L)
    dev = torch.device(DEV)
    x16 = torch.randn((sum(lengths), L), generator=torch.Generator().manual_seed(9)).to(DEV).to(torch.bfloat16)
    y = syn.make_labels(10, len(lengths)).to(DEV)
    lay = BagLayout.make(lengths, dev)
    a = ImageOnlyTrainer(p, dev, lr=1e-3)
    b = ImageOnlyTrainer(p, dev, lr=1e-3)
    for _ in range(3):
        la, _ = a.train_step(x16, lay, y)                      # one C call: ... -> fold with head gradients + Adam + shadows
        b.forward(x16, lay, y)
        b.backward()
        b.reduce_and_step()                                    # stand-alone k_adam + two k_cast_bf16
        assert float(la.item()) == float(b.loss_sum.item())
    torch.cuda.synchronize()
    assert torch.equal(a.fp.flat, b.fp.flat) and torch.equal(a.fp.exp_avg, b.fp.exp_avg) and torch.equal(a.fp.exp_avg_sq, b.fp.exp_avg_sq)
    assert torch.equal(a.fp.grad, b.fp.grad)
    for k in a._w16:
        assert torch.equal(a._w16[k].view(torch.int16), b._w16[k].view(torch.int16)), k
        assert torch.equal(a._w16[k], a.fp.p(k).to(torch.bfloat16)), k          # the shadow IS the rounded master


@pytest.mark.parametrize("train_mode", [False, True])
@pytest.mark.parametrize("lengths,L", [([9000, 7000, 424], 1024), ([16384], 512), ([5000, 3211], 1024)])
def test_eight_wave_weight_gradient_equals_the_four_wave_kernel(lengths, L, train_mode, monkeypatch):
    """Round 4: k_gate_bwd_dw_bf16_w8 (eight waves, 128 x 512 tiles, 32-row slices; MIL_DW16_W8=1) against the four-wave kernel
    (MIL_DW16_W8=0) on the same step: same bf16-rounded dPre and x, same fp32 accumulation, another split of the rows over
    workgroups - gradients agree to summation order (1e-5 rel), and both satisfy the oracle bar.  Row counts with partial
    last slices / chunks (16 424 = 513.25 slices of 32; 8 211 rows), one and two j tiles."""
    p = syn.image_only_params(5, L=L)
    dev = torch.device(DEV)
    x16 = torch.randn((sum(lengths), L), generator=torch.Generator().manual_seed(3)).to(DEV).to(torch.bfloat16)
    y = syn.make_labels(4, len(lengths)).to(DEV)
    lay = BagLayout.make(lengths, dev)
    out = {}
    for w8 in ("0", "1"):
        monkeypatch.setenv("MIL_DW16_W8", w8)
        tr = ImageOnlyTrainer(p, dev, lr=1e-3, train_mode=train_mode, seed=11)
        tr.forward(x16, lay, y)
        tr.backward()
        torch.cuda.synchronize()
        out[w8] = tr.fp.grad.clone()
        assert bool(torch.isfinite(out[w8]).all())
    from mil_amd.trainer import PARAM_ORDER
    for k in PARAM_ORDER:
        a = out["1"][tr.fp.offsets[k]:tr.fp.offsets[k] + tr.fp.p(k).numel()]
        b = out["0"][tr.fp.offsets[k]:tr.fp.offsets[k] + tr.fp.p(k).numel()]
        if k.endswith("attention_weights.bias"):
            continue
        assert rel_err(a.cpu(), b.cpu()) <= 1e-5, (k, rel_err(a.cpu(), b.cpu()))
