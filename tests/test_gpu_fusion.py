"""GPU parity of the cross-modal fusion kernels (K2: attention cores, LayerNorm, PE, two-way transformer) and
the CLIP text tower (K4) against the golden vectors of the reference's own modules and against the oracle."""
import pytest
import torch

from conftest import check_grad, load_golden, rel_err
from mil_amd import ops, synthetic as syn
from mil_amd.model.sam.transformer import Attention, TwoWayAttentionBlock, TwoWayTransformer
from mil_amd.clip.model import CLIPText
from oracle import mil_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 5e-5


def sub(p, prefix):
    return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}


@pytest.mark.parametrize("tag", ["attn_self_T10", "attn_t2i_T1_N64", "attn_t2i_T10_N128", "attn_i2t_N64_T10",
                                 "attn_i2t_N128_T1"])
def test_attention_module(tag):
    g = load_golden(tag)
    Tq, Tk, internal = [int(v) for v in g["shape"]]
    gen = torch.Generator().manual_seed(int(g["seed"]))
    p = {}
    syn.attention_params(p, gen, "attn", 512, internal)
    m = Attention(512, 8, downsample_rate=512 // internal).to(DEV)
    m.load_state_dict(sub(p, "attn."))
    q = torch.randn((1, Tq, 512), generator=gen).to(DEV).requires_grad_(True)
    k = torch.randn((1, Tk, 512), generator=gen).to(DEV).requires_grad_(True)
    v = torch.randn((1, Tk, 512), generator=gen).to(DEV).requires_grad_(True)
    out = m(q, k, v)
    go = torch.randn((1, Tq, 512), generator=gen).to(DEV)
    (out * go).sum().backward()
    assert rel_err(out[0].detach().cpu(), g["out"]) <= TOL
    assert rel_err(q.grad[0].cpu(), g["dq"]) <= TOL
    assert rel_err(k.grad[0].cpu(), g["dk"]) <= TOL
    assert rel_err(v.grad[0].cpu(), g["dv"]) <= TOL
    for n, t in m.named_parameters():
        if float(g["g.attn." + n + ".norm"]) > 0:
            check_grad("g.attn." + n, t.grad, g, 2e-4)


def test_layernorm_fwd_bwd():
    gen = torch.Generator().manual_seed(3)
    for rows, E in ((1, 512), (10, 512), (1000, 512), (77, 64)):
        x = torch.randn(rows, E, generator=gen) * 2 + 0.5
        w, b = torch.randn(E, generator=gen), torch.randn(E, generator=gen)
        go = torch.randn(rows, E, generator=gen)
        xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
        y = ops.layer_norm(xd, wd, bd)
        (y * go.to(DEV)).sum().backward()
        xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
        yr = torch.nn.functional.layer_norm(xr, (E,), wr, br, 1e-5)
        (yr * go).sum().backward()
        assert rel_err(y.detach().cpu(), yr.detach()) <= 1e-5
        assert rel_err(xd.grad.cpu(), xr.grad) <= 1e-4
        assert rel_err(wd.grad.cpu(), wr.grad) <= 1e-4 and rel_err(bd.grad.cpu(), br.grad) <= 1e-4


@pytest.mark.parametrize("frozen", [False, True])
def test_layernorm_with_residual_alias_and_frozen_parameters(frozen):
    """ops.layer_norm_res: pre-norm residual x + W LN(x); the skip gradient is added inside the LayerNorm backward kernel,
    frozen gamma / beta skip the parameter sums.  Row counts hit the one-workgroup kernel, odd pairs and the tiled one."""
    gen = torch.Generator().manual_seed(5)
    for rows, E in ((3, 512), (64, 512), (65, 512), (770, 512), (1001, 256)):
        x = torch.randn(rows, E, generator=gen)
        w, b = torch.randn(E, generator=gen), torch.randn(E, generator=gen)
        go = torch.randn(rows, E, generator=gen)
        xd = x.to(DEV).requires_grad_(True)
        wd, bd = (t.to(DEV).requires_grad_(not frozen) for t in (w, b))
        y, xr_ = ops.layer_norm_res(xd, wd, bd)
        ((torch.tanh(y) + xr_) * go.to(DEV)).sum().backward()
        xr = x.clone().requires_grad_(True)
        wr, br = (t.clone().requires_grad_(True) for t in (w, b))
        yr = torch.nn.functional.layer_norm(xr, (E,), wr, br, 1e-5)
        ((torch.tanh(yr) + xr) * go).sum().backward()
        assert rel_err(y.detach().cpu(), yr.detach()) <= 1e-5
        assert rel_err(xd.grad.cpu(), xr.grad) <= 1e-4, rows
        if frozen:
            assert wd.grad is None and bd.grad is None
        else:
            assert rel_err(wd.grad.cpu(), wr.grad) <= 1e-4 and rel_err(bd.grad.cpu(), br.grad) <= 1e-4


def test_device_pe_table_matches_reference_formula():
    pe = ops.sinusoid_pe(2000, 512, torch.device(DEV)).cpu()
    ref = orc.sinusoidal_pe(2000, 512)
    assert float((pe - ref).abs().max()) <= 2e-4          # fp32 exp/sin argument rounding at p ~ 2000
    assert float((pe[:64] - ref[:64]).abs().max()) <= 1e-5


@pytest.mark.parametrize("tag", ["twoway_block_skip", "twoway_block_noskip"])
def test_twoway_block(tag):
    g = load_golden(tag)
    full = syn.twoway_params(int(g["seed"]), "tw", depth=1)
    p = {k.replace("tw.layers.0.", ""): v for k, v in full.items() if k.startswith("tw.layers.0.")}
    m = TwoWayAttentionBlock(512, 8, 2048, "relu", 2, skip_first_layer_pe=bool(int(g["skip"]))).to(DEV)
    m.load_state_dict(p)
    with torch.no_grad():
        q, k = m(g["queries_in"][None].to(DEV), g["keys_in"][None].to(DEV), g["query_pe"][None].to(DEV),
                 g["key_pe"][None].to(DEV))
    assert rel_err(q[0].cpu(), g["queries"]) <= TOL and rel_err(k[0].cpu(), g["keys"]) <= TOL


@pytest.mark.parametrize("tag", ["twoway_T1_N64", "twoway_T10_N64", "twoway_T1_N200"])
def test_twoway_transformer(tag):
    g = load_golden(tag)
    seed = int(g["seed"])
    T, N = [int(v) for v in g["shape"]]
    name = "TwoWayTransformer_Pth"
    p = syn.twoway_params(seed, name)
    m = TwoWayTransformer(None, 2, 512, 8, 2048).to(DEV)
    m.load_state_dict(sub(p, name + "."))
    gen = torch.Generator().manual_seed(seed + 1)
    img = torch.randn((1, N, 512), generator=gen).to(DEV).requires_grad_(True)
    pt = torch.randn((1, T, 512), generator=gen).to(DEV).requires_grad_(True)
    pe = orc.sinusoidal_pe(N, 512)[None].to(DEV)
    q, k = m(img, pe, pt)
    gq = torch.randn((1, T, 512), generator=gen).to(DEV)
    gk = torch.randn((1, N, 512), generator=gen).to(DEV)
    ((q * gq).sum() + (k * gk).sum()).backward()
    assert rel_err(q[0].detach().cpu(), g["queries"]) <= TOL and rel_err(k[0].detach().cpu(), g["keys"]) <= TOL
    assert rel_err(img.grad[0].cpu(), g["dimage"]) <= 2e-4 and rel_err(pt.grad[0].cpu(), g["dpoint"]) <= 2e-4
    for n, t in m.named_parameters():
        gn = float(g["g." + name + "." + n + ".norm"])
        got = t.grad if t.grad is not None else torch.zeros_like(t)
        if gn == 0.0:
            assert float(got.abs().max()) <= 1e-10, n
        else:
            check_grad("g." + name + "." + n, got, g, 5e-4)


def test_twoway_ragged_batch_equals_per_bag():
    """Two bags of different size in one flat launch sequence == each bag alone."""
    name = "TwoWayTransformer_Pth"
    p = syn.twoway_params(5, name)
    m = TwoWayTransformer(None, 2, 512, 8, 2048).to(DEV)
    m.load_state_dict(sub(p, name + "."))
    gen = torch.Generator().manual_seed(6)
    ns, ts = [70, 129], [1, 1]
    imgs = [torch.randn(n, 512, generator=gen).to(DEV) for n in ns]
    pts = [torch.randn(t, 512, generator=gen).to(DEV) for t in ts]
    pe = orc.sinusoidal_pe(max(ns), 512).to(DEV)
    with torch.no_grad():
        q, k = m.flat(torch.cat(imgs), torch.cat(pts), pe, ns, ts)
        for i in range(2):
            qi, ki = m.flat(imgs[i], pts[i], pe, [ns[i]], [ts[i]])
            assert rel_err(q[i:i + 1], qi) <= 1e-6
            assert rel_err(k[sum(ns[:i]):sum(ns[:i + 1])], ki) <= 1e-6


@pytest.mark.parametrize("tag", ["clip_text_small", "clip_text_vitb32"])
def test_clip_text_tower(tag):
    g = load_golden(tag)
    width, layers, vocab, heads, embed, P = [int(v) for v in g["cfg"]]
    p = syn.clip_text_params(int(g["seed"]), width=width, layers=layers, vocab=vocab, embed=embed)
    m = CLIPText(embed, 77, vocab, width, heads, layers)
    m.load_state_dict(sub(p, "clinic_extractor.model."), strict=False)
    m = m.to(DEV)
    out = m.encode_text(g["ids"].to(DEV))
    assert rel_err(out.cpu(), g["out"]) <= TOL
