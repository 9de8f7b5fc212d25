"""GPU parity: the HIP image-only path (gate scores -> attention pool -> head -> BCE -> backward)
through the C ABI, against the CPU oracle on the same seeded inputs and against the golden vectors
produced by the reference's leaf modules.  Bars: top-1 bit-exact, |logits| <= 1e-3 (fp32 path; we
assert a tighter 2e-5), gradient rel-err <= 1e-3 (asserted 2e-4)."""
import pytest
import torch

from conftest import check_grad, load_golden, rel_err
from mil_amd import ops, synthetic as syn
from mil_amd.bags import BagLayout
from oracle import mil_oracle as orc

pytestmark = pytest.mark.gpu
LOGIT_TOL = 2e-5
GRAD_TOL = 2e-4


def _bags(seed, lengths, L):
    return [torch.randn((n, L), generator=torch.Generator().manual_seed(seed + 100 + i))
            for i, n in enumerate(lengths)]


def hip_image_only(xcat, p, lengths, y=None, want_dx=False):
    dev = torch.device("cuda")
    d = {k: v.to(dev) for k, v in p.items()}
    x = xcat.to(dev)
    lay = BagLayout.make(lengths, dev)
    Wv, bv = d["aggregator.attention_V.0.weight"], d["aggregator.attention_V.0.bias"]
    Wu, bu = d["aggregator.attention_U.0.weight"], d["aggregator.attention_U.0.bias"]
    w, b = d["aggregator.attention_weights.weight"].reshape(-1), d["aggregator.attention_weights.bias"]
    Wf, bf = d["fc.1.weight"], d["fc.1.bias"]
    scores, gates = ops.gate_scores_fwd(x, Wv, bv, Wu, bu, w, b, save_gates=True)
    M, lse = ops.attn_pool_fwd(x, scores, lay)
    z, prob = ops.head_fwd(M, Wf, bf)
    out = {"scores": scores, "M": M, "lse": lse, "logits": z, "prob": prob}
    if y is not None:
        B, C = prob.shape
        loss, dz = ops.bce_fwd_bwd(prob, y.to(dev), 1.0 / (B * C))
        dM, dWf, dbf, cdot = ops.head_bwd(dz, None, M, Wf)
        ds, dx = ops.attn_pool_bwd(x, scores, lse, dM, cdot, lay, want_dx)
        g = {k: torch.empty_like(v) for k, v in d.items() if k.startswith("aggregator.")}
        ops.gate_bwd_params(x, gates, ds, w, g["aggregator.attention_V.0.weight"], g["aggregator.attention_V.0.bias"],
                            g["aggregator.attention_U.0.weight"], g["aggregator.attention_U.0.bias"],
                            g["aggregator.attention_weights.weight"].view(-1), g["aggregator.attention_weights.bias"])
        if want_dx:
            ops.gate_bwd_input(gates, ds, w, Wv, Wu, dx)
        g["fc.1.weight"], g["fc.1.bias"] = dWf, dbf
        out.update(loss=loss, grads=g, dx=dx, ds=ds)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("tag,L", [("image_only_n7", 512), ("image_only_8x128", 512), ("image_only_ragged", 512),
                                   ("image_only_4x1024", 512), ("image_only_2x4096_L1024", 1024)])
def test_image_only_vs_golden_and_oracle(tag, L):
    g = load_golden(tag)
    seed = int(g["seed"])
    p = syn.image_only_params(seed, L=L)
    lengths = [int(v) for v in g["lengths"]]
    bags = _bags(seed, lengths, L)
    y = syn.make_labels(seed + 7, len(lengths))
    out = hip_image_only(torch.cat(bags, 0), p, lengths, y, want_dx=True)
    # --- golden vectors (reference leaf modules)
    assert float((out["logits"].cpu() - g["logits"]).abs().max()) <= LOGIT_TOL
    assert torch.equal(out["prob"].cpu().argmax(-1), g["prob"].argmax(-1))            # top-1 bit-exact
    assert abs(float(out["loss"].cpu()) - float(g["loss"])) <= 1e-5
    assert float((out["M"].cpu() - g["M"]).abs().max()) <= 1e-5
    assert float((out["scores"].cpu() - g["scores"]).abs().max()) <= 1e-5
    for k, v in out["grads"].items():
        check_grad("g." + k, v.reshape(p[k].shape), g, GRAD_TOL)
    check_grad("dx", out["dx"], g, GRAD_TOL)
    # --- oracle on the same inputs (full tensors)
    loss, logits, prob, grads = orc.batch_loss_and_grads(bags, y, p)
    assert float((out["logits"].cpu() - logits).abs().max()) <= LOGIT_TOL
    assert torch.equal(out["prob"].cpu().argmax(-1), prob.argmax(-1))
    for k, v in out["grads"].items():
        if float(grads[k].norm()) > 1e-7:
            assert rel_err(v.cpu().reshape(grads[k].shape), grads[k]) <= GRAD_TOL, k


def test_single_row_bag_and_tail_tiles():
    """Edge cases: a bag of one patch (softmax weight exactly 1), R not a multiple of the 128-row
    gate tile or the 32-row pool tile."""
    L = 512
    p = syn.image_only_params(5, L=L)
    lengths = [1, 33, 129]
    bags = _bags(5, lengths, L)
    out = hip_image_only(torch.cat(bags, 0), p, lengths)
    assert float((out["M"][0].cpu() - bags[0][0]).abs().max()) == 0.0
    for i, xb in enumerate(bags):
        o = orc.image_only_forward(xb, p)
        assert float((out["logits"][i].cpu() - o["logits"][0]).abs().max()) <= LOGIT_TOL


def test_softmax_shift_invariance_large_scores():
    """Size-independent property: adding a constant to the attention bias leaves M unchanged, also
    when scores are large (online-softmax max subtraction across tiles)."""
    L = 512
    p = syn.image_only_params(9, L=L)
    lengths = [200, 77]
    x = torch.cat(_bags(9, lengths, L), 0)
    a = hip_image_only(x, p, lengths)
    p2 = dict(p)
    p2["aggregator.attention_weights.bias"] = p["aggregator.attention_weights.bias"] + 80.0
    b = hip_image_only(x, p2, lengths)
    assert float((a["M"] - b["M"]).abs().max()) <= 1e-5
    assert float(((b["lse"] - a["lse"]) - 80.0).abs().max()) <= 1e-4


def test_spiked_score_forces_cross_tile_rescale():
    """One row with a dominant score in the LAST tile: partial merge must rescale earlier tiles."""
    L = 512
    p = syn.image_only_params(13, L=L)
    lengths = [100]
    x = _bags(13, lengths, L)[0]
    wv = p["aggregator.attention_V.0.weight"]
    x[97] = 40.0 * wv[0] / wv[0].norm()        # drives one gate unit into saturation
    out = hip_image_only(x, p, lengths)
    o = orc.image_only_forward(x, p)
    assert float((out["M"].cpu() - o["M"]).abs().max()) <= 2e-5
    assert float((out["logits"].cpu() - o["logits"]).abs().max()) <= LOGIT_TOL


def test_full_size_properties_config2():
    """BASELINE config 2 size (32 x 1024 x 512): properties that need no CPU reference at scale:
    pool weights sum to 1 (M of a constant bag is the constant), permutation invariance over the
    patches of a bag, and bags are independent."""
    dev = torch.device("cuda")
    B, N, L = 32, 1024, 512
    p = syn.image_only_params(1234, L=L)
    x = syn.make_bags(4321, B, N, L)
    out = hip_image_only(x.reshape(B * N, L), p, [N] * B)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(1))
    xp = x.clone()
    xp[3] = x[3][perm]
    outp = hip_image_only(xp.reshape(B * N, L), p, [N] * B)
    assert float((out["logits"] - outp["logits"]).abs().max()) <= 2e-6
    single = hip_image_only(x[7], p, [N])
    assert float((single["logits"][0] - out["logits"][7]).abs().max()) <= 1e-6
    const = torch.ones(N, L) * 0.37
    oc = hip_image_only(const, p, [N])
    assert float((oc["M"].cpu() - 0.37).abs().max()) <= 1e-6
    # oracle on 2 of the 32 bags
    for b in (0, 31):
        o = orc.image_only_forward(x[b], p)
        assert float((out["logits"][b].cpu() - o["logits"][0]).abs().max()) <= LOGIT_TOL


def test_autograd_wrappers_match_plain_ops():
    dev = torch.device("cuda")
    L = 512
    p = {k: v.to(dev).requires_grad_(True) for k, v in syn.image_only_params(3, L=L).items()}
    lengths = [50, 70]
    x = torch.cat(_bags(3, lengths, L), 0).to(dev).requires_grad_(True)
    y = syn.make_labels(10, 2).to(dev)
    lay = BagLayout.make(lengths, dev)
    M, _ = ops.gated_attention_pool(x, p["aggregator.attention_V.0.weight"], p["aggregator.attention_V.0.bias"],
                                    p["aggregator.attention_U.0.weight"], p["aggregator.attention_U.0.bias"],
                                    p["aggregator.attention_weights.weight"], p["aggregator.attention_weights.bias"], lay)
    prob, z = ops.head_sigmoid(M, p["fc.1.weight"], p["fc.1.bias"])
    loss = torch.nn.BCELoss()(prob, y)
    loss.backward()
    cpu = {k: v.detach().cpu() for k, v in p.items()}
    bags = list(x.detach().cpu().split(lengths))
    l2, _, _, grads = orc.batch_loss_and_grads(bags, y.cpu(), cpu)
    assert abs(float(loss) - float(l2)) <= 1e-5
    for k, v in p.items():
        if float(grads[k].norm()) > 1e-7:
            assert rel_err(v.grad.cpu(), grads[k]) <= GRAD_TOL, k
