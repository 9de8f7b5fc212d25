"""CPU: the oracle restatement (oracle/mil_oracle.py) against the golden vectors produced by
the reference's own leaf modules (oracle/gen_golden.py, run in the build container)."""
import pytest
import torch

from conftest import check_grad, load_golden, rel_err
from mil_amd import synthetic as syn
from oracle import mil_oracle as orc

TOL = 2e-5


def _bags(seed, lengths, L):
    return [torch.randn((n, L), generator=torch.Generator().manual_seed(seed + 100 + i))
            for i, n in enumerate(lengths)]


@pytest.mark.parametrize("tag,L", [("image_only_n7", 512), ("image_only_8x128", 512), ("image_only_ragged", 512),
                                   ("image_only_4x1024", 512), ("image_only_2x4096_L1024", 1024)])
def test_image_only(tag, L):
    g = load_golden(tag)
    seed = int(g["seed"])
    p = syn.image_only_params(seed, L=L)
    lengths = [int(v) for v in g["lengths"]]
    bags = _bags(seed, lengths, L)
    if "x" in g:
        assert torch.equal(torch.cat(bags, 0), g["x"])
        for k, v in p.items():
            assert torch.equal(v, g["p." + k]), k
    y = syn.make_labels(seed + 7, len(lengths))
    assert torch.equal(y, g["labels"])
    xs = [b.clone().requires_grad_(True) for b in bags]
    names = list(p.keys())
    leaves = {k: p[k].clone().requires_grad_(True) for k in names}
    outs = [orc.image_only_forward(x, leaves) for x in xs]
    prob = torch.cat([o["prob"] for o in outs], 0)
    logits = torch.cat([o["logits"] for o in outs], 0)
    loss = orc.bce_loss(prob, y)
    loss.backward()
    assert float((logits.detach() - g["logits"]).abs().max()) <= 1e-6
    assert torch.equal(orc.top1(prob), g["prob"].argmax(-1))
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-6
    assert float((torch.cat([o["M"] for o in outs], 0) - g["M"]).abs().max()) <= 1e-6
    assert float((torch.cat([o["scores"] for o in outs], 0) - g["scores"]).abs().max()) <= 1e-6
    for k in names:
        check_grad("g." + k, leaves[k].grad, g, TOL)
    check_grad("dx", torch.cat([x.grad for x in xs], 0), g, TOL)


def test_batched_quirk_documented():
    g = load_golden("abmil_batched_quirk")
    p = syn.image_only_params(int(g["seed"]), L=512)
    out = orc.abmil_forward_batched_quirk(g["x"], p)
    assert rel_err(out, g["out"]) <= 1e-6
    assert rel_err(out, g["x"].sum(1, keepdim=True)) <= 1e-6      # the "pool" is a plain sum over N


@pytest.mark.parametrize("tag", ["attn_self_T10", "attn_t2i_T1_N64", "attn_t2i_T10_N128", "attn_i2t_N64_T10",
                                 "attn_i2t_N128_T1"])
def test_attention(tag):
    g = load_golden(tag)
    Tq, Tk, internal = [int(v) for v in g["shape"]]
    gen = torch.Generator().manual_seed(int(g["seed"]))
    p = {}
    syn.attention_params(p, gen, "attn", 512, internal)
    q = torch.randn((1, Tq, 512), generator=gen)[0].requires_grad_(True)
    k = torch.randn((1, Tk, 512), generator=gen)[0].requires_grad_(True)
    v = torch.randn((1, Tk, 512), generator=gen)[0].requires_grad_(True)
    leaves = {n: t.clone().requires_grad_(True) for n, t in p.items()}
    out = orc.attention(q, k, v, leaves, "attn")
    go = torch.randn((1, Tq, 512), generator=gen)[0]
    (out * go).sum().backward()
    assert rel_err(out, g["out"]) <= TOL
    assert rel_err(q.grad, g["dq"]) <= TOL and rel_err(k.grad, g["dk"]) <= TOL and rel_err(v.grad, g["dv"]) <= TOL
    for n in p:
        if leaves[n].grad is not None and float(g["g." + n + ".norm"]) > 0:
            check_grad("g." + n, leaves[n].grad, g, 1e-4)


@pytest.mark.parametrize("tag", ["twoway_block_skip", "twoway_block_noskip"])
def test_twoway_block(tag):
    g = load_golden(tag)
    full = syn.twoway_params(int(g["seed"]), "tw", depth=1)
    p = {k.replace("tw.layers.0.", "blk."): v for k, v in full.items() if k.startswith("tw.layers.0.")}
    q, k = orc.twoway_block(g["queries_in"], g["keys_in"], g["query_pe"], g["key_pe"], p, "blk",
                            skip_first_layer_pe=bool(int(g["skip"])))
    assert rel_err(q, g["queries"]) <= TOL and rel_err(k, g["keys"]) <= TOL


@pytest.mark.parametrize("tag", ["twoway_T1_N64", "twoway_T10_N64", "twoway_T1_N200"])
def test_twoway(tag):
    g = load_golden(tag)
    seed = int(g["seed"])
    T, N = [int(v) for v in g["shape"]]
    name = "TwoWayTransformer_Pth"
    p = syn.twoway_params(seed, name)
    gen = torch.Generator().manual_seed(seed + 1)
    img = torch.randn((1, N, 512), generator=gen)[0].requires_grad_(True)
    pt = torch.randn((1, T, 512), generator=gen)[0].requires_grad_(True)
    leaves = {n: t.clone().requires_grad_(True) for n, t in p.items()}
    q, k = orc.twoway_transformer(img, orc.sinusoidal_pe(N, 512), pt, leaves, name)
    gq = torch.randn((1, T, 512), generator=gen)[0]
    gk = torch.randn((1, N, 512), generator=gen)[0]
    ((q * gq).sum() + (k * gk).sum()).backward()
    assert rel_err(q, g["queries"]) <= TOL and rel_err(k, g["keys"]) <= TOL
    assert rel_err(img.grad, g["dimage"]) <= 1e-4 and rel_err(pt.grad, g["dpoint"]) <= 1e-4
    for n in p:
        gn = float(g["g." + n + ".norm"])
        got = leaves[n].grad if leaves[n].grad is not None else torch.zeros_like(leaves[n])
        if gn == 0.0:       # T=1: q/k proj of image->token attention get exactly zero gradient
            assert float(got.abs().max()) <= 1e-12, n
        else:
            check_grad("g." + n, got, g, 2e-4)


@pytest.mark.parametrize("tag", ["clip_text_small", "clip_text_vitb32"])
def test_clip_text(tag):
    g = load_golden(tag)
    width, layers, vocab, heads, embed, P = [int(v) for v in g["cfg"]]
    p = syn.clip_text_params(int(g["seed"]), width=width, layers=layers, vocab=vocab, embed=embed)
    ids = syn.make_token_ids(int(g["seed"]) + 1, 1, P, vocab=vocab)[0]
    assert torch.equal(ids, g["ids"])
    out = orc.clip_encode_text(ids, p, heads)
    assert rel_err(out, g["out"]) <= TOL


@pytest.mark.parametrize("tag", ["fused_small_clip", "fused_P10", "fused_vitb32"])
def test_fused(tag):
    g = load_golden(tag)
    seed = int(g["seed"])
    B, N, P, clayers, cwidth, cvocab, cheads = [int(v) for v in g["cfg"]]
    name = "TwoWayTransformer_Pth"
    p = syn.fused_params(seed, name, clip_width=cwidth, clip_layers=clayers, clip_vocab=cvocab)
    x = syn.make_bags(seed + 3, B, N, 768)
    ids = syn.make_token_ids(seed + 4, B, P, vocab=cvocab)
    y = syn.make_labels(seed + 5, B)
    train = [k for k in p if not k.startswith("clinic_extractor.")]
    leaves = dict(p)
    leaves.update({k: p[k].clone().requires_grad_(True) for k in train})
    outs = [orc.fused_forward(x[b], ids[b], leaves, cheads, name) for b in range(B)]
    prob = torch.cat([o["prob"] for o in outs], 0)
    logits = torch.cat([o["logits"] for o in outs], 0)
    loss = orc.bce_loss(prob, y)
    loss.backward()
    assert float((logits - g["logits"]).abs().max()) <= 2e-6
    assert abs(float(loss) - float(g["loss"])) <= 1e-6
    assert rel_err(torch.stack([o["text"] for o in outs], 0), g["text"]) <= TOL
    assert rel_err(torch.stack([o["x_Pth2CI"] for o in outs], 0), g["x_Pth2CI"]) <= TOL
    for k in train:
        gn = float(g["g." + k + ".norm"])
        got = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])
        if gn == 0.0:
            assert float(got.abs().max()) <= 1e-12, k
        else:
            check_grad("g." + k, got, g, 5e-4)


def test_pe_against_float64():
    import numpy as np
    pe = orc.sinusoidal_pe(300, 512).numpy()
    pos = np.arange(300, dtype=np.float64)[:, None]
    div = np.exp(np.arange(0, 512, 2, dtype=np.float64) * -(np.log(10000.0) / 512))
    assert np.abs(pe[:, 0::2] - np.sin(pos * div)).max() < 1e-4      # fp32 argument rounding at p~300
    assert np.abs(pe[:, 1::2] - np.cos(pos * div)).max() < 1e-4
    import math
    div32 = torch.exp(torch.arange(0, 512, 2, dtype=torch.float) * -(math.log(10000.0) / 512))
    arg32 = (torch.arange(0, 300).unsqueeze(1).float() * div32).double().numpy()    # the fp32 argument, exactly
    assert np.abs(pe[:, 0::2] - np.sin(arg32)).max() < 2e-6
    assert pe[0, 0] == 0.0 and pe[0, 1] == 1.0


def test_sampler_matches_torch():
    from torch.utils.data import DistributedSampler

    class _DS:
        def __len__(self):
            return 37

    for world in (1, 2, 4, 8):
        for epoch in (0, 1):
            for rank in range(world):
                s = DistributedSampler(_DS(), num_replicas=world, rank=rank, shuffle=True)
                s.set_epoch(epoch)
                assert list(s) == orc.distributed_sampler_indices(37, world, rank, epoch)


def test_adam_matches_torch():
    torch.manual_seed(0)
    w = torch.randn(50)
    ref = w.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-5, betas=(0.9, 0.999), weight_decay=1e-7)
    m = torch.zeros(50); v = torch.zeros(50); cur = w.clone()
    for step in range(1, 4):
        gr = torch.randn(50)
        ref.grad = gr.clone()
        opt.step()
        cur, m, v = orc.adam_step(cur, gr, m, v, step)
        assert float((cur - ref.detach()).abs().max()) < 1e-7


@pytest.mark.parametrize("tag", ["coop_small", "coop_w512"])
def test_learnable_prompts(tag):
    g = load_golden(tag)
    width, layers, vocab, heads, embed, P, n_ctx = [int(v) for v in g["cfg"]]
    seed = int(g["seed"])
    p = syn.clip_text_params(seed, width=width, layers=layers, vocab=vocab, embed=embed)
    ids = syn.make_token_ids(seed + 1, 1, P, vocab=vocab)[0]
    gen = torch.Generator().manual_seed(seed + 2)
    ctx = (torch.randn((P, n_ctx, width), generator=gen) * 0.02).requires_grad_(True)
    out = orc.clip_learnable_prompts(ids, ctx, p, heads)
    go = torch.randn(out.shape, generator=gen)
    (out * go).sum().backward()
    assert rel_err(out.detach(), g["out"]) <= TOL
    assert rel_err(ctx.grad, g["dctx"]) <= 1e-4


@pytest.mark.parametrize("tag", ["twoway_ctmap_T1", "twoway_ctmap_T10"])
def test_twoway_on_a_ct_map(tag):
    """The 5-D branch of sam/transformer.py:78-98 (CT feature map -> 160 tokens) against the reference's own forward."""
    g = load_golden(tag)
    seed = int(g["seed"])
    T, D, hw = [int(v) for v in g["shape"]]
    name = "TwoWayTransformer_CT"
    p = syn.twoway_params(seed, name)
    ct = syn.make_ct_map(seed + 1, 1, D, hw)
    gen = torch.Generator().manual_seed(seed + 2)
    pt = torch.randn((1, T, 512), generator=gen)[0].requires_grad_(True)
    leaves = {n: t.clone().requires_grad_(True) for n, t in p.items()}
    q, k = orc.twoway_transformer(orc.ct_map_tokens(ct)[0], orc.sinusoidal_pe(D, 512), pt, leaves, name)
    gq = torch.randn((1, T, 512), generator=gen)[0]
    gk = torch.randn((1, D, 512), generator=gen)[0]
    ((q * gq).sum() + (k * gk).sum()).backward()
    assert rel_err(q, g["queries"]) <= TOL and rel_err(k, g["keys"]) <= TOL
    assert rel_err(pt.grad, g["dpoint"]) <= 1e-4
    for n in p:
        gn = float(g["g." + n + ".norm"])
        got = leaves[n].grad if leaves[n].grad is not None else torch.zeros_like(leaves[n])
        if gn == 0.0:
            assert float(got.abs().max()) <= 1e-12, n
        else:
            check_grad("g." + n, got, g, 2e-4)


@pytest.mark.parametrize("tag", ["twoway_ctbase_N64", "twoway_ctbase_D160"])
def test_twoway_alignment_base_ct(tag):
    """`--alignment_base CT` (sam/transformer.py:78-86): the CT map's D tokens are the QUERIES (point embedding) of the two-way
    transformer, the image tokens the keys - the oracle restatement against the reference's own forward."""
    g = load_golden(tag)
    seed = int(g["seed"])
    N, D, hw = [int(v) for v in g["shape"]]
    name = "TwoWayTransformer_Pth"
    p = syn.twoway_params(seed, name)
    ct = syn.make_ct_map(seed + 1, 1, D, hw)
    gen = torch.Generator().manual_seed(seed + 2)
    img = torch.randn((1, N, 512), generator=gen)[0].requires_grad_(True)
    leaves = {n: t.clone().requires_grad_(True) for n, t in p.items()}
    q, k = orc.twoway_transformer(img, orc.sinusoidal_pe(N, 512), orc.ct_map_tokens(ct)[0], leaves, name)
    gq = torch.randn((1, D, 512), generator=gen)[0]
    gk = torch.randn((1, N, 512), generator=gen)[0]
    ((q * gq).sum() + (k * gk).sum()).backward()
    assert rel_err(q, g["queries"]) <= TOL and rel_err(k, g["keys"]) <= TOL
    assert rel_err(img.grad, g["dimage"]) <= 1e-4
    for n in p:
        gn = float(g["g." + n + ".norm"])
        got = leaves[n].grad if leaves[n].grad is not None else torch.zeros_like(leaves[n])
        if gn == 0.0:
            assert float(got.abs().max()) <= 1e-12, n
        else:
            check_grad("g." + n, got, g, 2e-4)


def test_fused_ct_and_pathology():
    """modality ['CT', 'pathology'] with a precomputed CT map: TwoWayTransformer_Both twice, 4-segment bag (aggregator.py:155-173)."""
    g = load_golden("fused_ct_pth")
    seed = int(g["seed"])
    B, N, P, D, hw, clayers = [int(v) for v in g["cfg"]]
    p = syn.fused_params(seed, "TwoWayTransformer_Both", clip_layers=clayers, with_ct=True)
    x = syn.make_bags(seed + 3, B, N, 768)
    ids = syn.make_token_ids(seed + 4, B, P)
    y = syn.make_labels(seed + 5, B)
    ct = syn.make_ct_map(seed + 6, B, D, hw)
    train = [k for k in p if not k.startswith("clinic_extractor.")]
    leaves = dict(p)
    leaves.update({k: p[k].clone().requires_grad_(True) for k in train})
    outs = [orc.fused_forward_ct_pth(ct[b], x[b], ids[b], leaves) for b in range(B)]
    prob = torch.cat([o["prob"] for o in outs], 0)
    logits = torch.cat([o["logits"] for o in outs], 0)
    loss = orc.bce_loss(prob, y)
    loss.backward()
    assert float((logits - g["logits"]).abs().max()) <= 2e-6
    assert abs(float(loss) - float(g["loss"])) <= 1e-6
    assert rel_err(torch.stack([o["x_CT2CI"] for o in outs], 0), g["x_CT2CI"]) <= TOL
    assert rel_err(torch.stack([o["x_Pth2CI"] for o in outs], 0), g["x_Pth2CI"]) <= TOL
    for k in train:
        gn = float(g["g." + k + ".norm"])
        got = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])
        if gn == 0.0:
            assert float(got.abs().max()) <= 1e-12, k
        else:
            check_grad("g." + k, got, g, 5e-4)


@pytest.mark.parametrize("case", ["tok", "rnd"])
def test_cosine_embedding_loss_restatement_vs_torch(golden, case):
    """The 'textCosSim' term (reference train_ddp.py:102,325-329 calls torch.nn.CosineEmbeddingLoss): restatement vs the
    vectors torch's own op produced on the fused_ct_pth tokens (oracle/gen_golden.py cossim)."""
    g = golden("cossim_ct_pth")
    a, b = g[case + ".x1"].clone().requires_grad_(True), g[case + ".x2"].clone().requires_grad_(True)
    loss = orc.cosine_embedding_loss(a, b)
    loss.backward()
    assert abs(float(loss) - float(g[case + ".loss"])) <= 1e-6
    assert rel_err(a.grad, g[case + ".dx1"]) <= 1e-5 and rel_err(b.grad, g[case + ".dx2"]) <= 1e-5
