"""GPU: LayerNorm(keys + row[bag]) fused with the next attention site's absorbed pool (ops.lnbr_one_token_attention:
mil_lnbr_absorbed_pool_value_fwd / _bwd) against the two separate nodes it replaces (ops.layer_norm_bag_row followed by
ops.one_token_attention - themselves pinned to the reference's TwoWayAttentionBlock by tests/golden/twoway_*.npz) and
against torch on the materialised tensors: outputs, and the gradient of every input."""
import os

import pytest
import torch

from conftest import rel_err
from mil_amd import ops
from mil_amd.segments import AttnSegs, FusionBucket

pytestmark = pytest.mark.gpu
DEV = "cuda"
E, H = 512, 8


def _inputs(lengths, C, seed=11):
    g = torch.Generator().manual_seed(seed)
    rows, B = sum(lengths), len(lengths)
    r = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(DEV)        # noqa: E731
    t = dict(x=r(rows, E), row=r(B, E), gamma=1 + r(E, sc=0.1), beta=r(E, sc=0.1), qp=r(B, H * C),
             Wk=r(H * C, E, sc=0.05), Wv=r(H * C, E, sc=0.05), bv=r(H * C, sc=0.05))
    pe = r(max(lengths) + 3, E, sc=0.3)
    do = r(B, H * C)
    dy = r(rows, E)
    return t, pe, do, dy


def _run(fused: bool, t, pe, do, dy, segs_ti, segs_it, tail_rows=0):
    leaves = {k: v.clone().requires_grad_(True) for k, v in t.items()}
    if fused:
        o, keys = ops.lnbr_one_token_attention(leaves["x"], leaves["row"], leaves["gamma"], leaves["beta"], 1e-5, pe, segs_ti,
                                               leaves["Wk"], leaves["Wv"], leaves["bv"], H, leaves["qp"], tail_rows)
    else:
        keys0 = ops.layer_norm_bag_row(leaves["x"], leaves["row"], segs_it, leaves["gamma"], leaves["beta"], 1e-5, tail_rows)
        o, keys = ops.one_token_attention(None, keys0, pe, segs_ti, None, None, leaves["Wk"], leaves["Wv"], leaves["bv"], H,
                                          qp=leaves["qp"])
    n = dy.shape[0]
    ((o * do).sum() + (keys[:n] * dy).sum()).backward()
    return o.detach(), keys.detach(), {k: v.grad for k, v in leaves.items()}


@pytest.mark.parametrize("lengths,C", [([300, 77, 512, 129], 32), ([1024] * 4, 32), ([640, 65, 2000], 64), ([70], 32)])
def test_fused_node_matches_the_two_nodes(lengths, C):
    t, pe, do, dy = _inputs(lengths, C)
    B = len(lengths)
    dev = torch.device(DEV)
    s_ti = AttnSegs.make([1] * B, lengths, dev)
    s_it = AttnSegs.make(lengths, [1] * B, dev)
    o1, k1, g1 = _run(True, t, pe, do, dy, s_ti, s_it, tail_rows=B)
    o0, k0, g0 = _run(False, t, pe, do, dy, s_ti, s_it, tail_rows=B)
    assert float((k1 - k0).abs().max()) <= 2e-6
    assert rel_err(o1, o0) <= 2e-6
    for name in g0:
        assert g1[name] is not None, name
        assert rel_err(g1[name], g0[name]) <= 2e-5, (name, rel_err(g1[name], g0[name]))


def test_three_kernel_backward_matches_the_one_pass_backward(monkeypatch):
    """MIL_LNBR_BWD=r16: per-row dots + dQp partials + LayerNorm backward with the rank-16 update in its load, the form the
    one-pass kernel (default) replaced."""
    lengths, C = [300, 77, 512, 129], 32
    t, pe, do, dy = _inputs(lengths, C, seed=21)
    B = len(lengths)
    s_ti = AttnSegs.make([1] * B, lengths, torch.device(DEV))
    _, _, g1 = _run(True, t, pe, do, dy, s_ti, None)
    monkeypatch.setenv("MIL_LNBR_BWD", "r16")
    _, _, g0 = _run(True, t, pe, do, dy, s_ti, None)
    for name in g0:
        assert rel_err(g1[name], g0[name]) <= 1e-5, (name, rel_err(g1[name], g0[name]))


@pytest.mark.parametrize("C", [32, 64])
def test_fused_node_matches_torch_on_materialised_tensors(C):
    lengths = [200, 333, 64]
    t, pe, do, dy = _inputs(lengths, C, seed=5)
    B = len(lengths)
    dev = torch.device(DEV)
    s_ti = AttnSegs.make([1] * B, lengths, dev)
    o1, k1, g1 = _run(True, t, pe, do, dy, s_ti, None)
    lv = {k: v.clone().requires_grad_(True) for k, v in t.items()}
    bag = torch.repeat_interleave(torch.arange(B, device=DEV), torch.tensor(lengths, device=DEV))
    keys = torch.nn.functional.layer_norm(lv["x"] + lv["row"][bag], (E,), lv["gamma"], lv["beta"], 1e-5)
    off = [0]
    for n in lengths:
        off.append(off[-1] + n)
    outs = []
    for b, n in enumerate(lengths):                        # the reference's Attention with q/k/v projected (transformer.py:428-450)
        kb = keys[off[b]:off[b + 1]]
        kp = (kb + pe[:n]) @ lv["Wk"].t()
        vp = kb @ lv["Wv"].t() + lv["bv"]
        q = lv["qp"][b].view(H, 1, C)
        a = torch.softmax(q @ kp.view(n, H, C).permute(1, 2, 0) / C ** 0.5, dim=-1)
        outs.append((a @ vp.view(n, H, C).permute(1, 0, 2)).reshape(H * C))
    o_ref = torch.stack(outs)
    ((o_ref * do).sum() + (keys * dy).sum()).backward()
    assert rel_err(o1, o_ref.detach()) <= 5e-6
    assert float((k1 - keys.detach()).abs().max()) <= 2e-5
    for name in t:
        assert rel_err(g1[name], lv[name].grad) <= 5e-5, (name, rel_err(g1[name], lv[name].grad))


def test_fused_node_on_a_capacity_bucket_padding_rows_are_zero():
    """Device-side lengths (segments.FusionBucket): rows beyond the bags get zero keys and exactly zero gradient, the
    real rows the numbers of the exact-shape run."""
    lengths, C, cap = [700, 420], 32, 2048
    t, pe, do, dy = _inputs(lengths, C, seed=3)
    B, n = len(lengths), sum(lengths)
    dev = torch.device(DEV)
    pe = torch.cat([pe, torch.zeros(cap, E, device=DEV)])[:cap].contiguous()
    o0, k0, g0 = _run(True, t, pe, do, dy, AttnSegs.make([1] * B, lengths, dev), None)
    bucket = FusionBucket(cap, B, dev, 1)
    bucket.set_lengths(lengths)
    bucket.refresh()
    tb = dict(t)
    tb["x"] = torch.cat([t["x"], torch.randn(cap - n, E, device=DEV)])        # junk in the padding rows
    dyb = torch.cat([dy, torch.randn(cap - n, E, device=DEV)])
    o1, k1, g1 = _run(True, tb, pe, do, dyb, bucket.s_ti, None)
    assert rel_err(o1, o0) <= 1e-6
    assert float((k1[:n] - k0).abs().max()) <= 1e-6
    assert float(k1[n:].abs().max()) == 0.0
    assert float(g1["x"][n:].abs().max()) == 0.0
    assert rel_err(g1["x"][:n], g0["x"]) <= 1e-5
    for name in ("row", "gamma", "beta", "qp", "Wk", "Wv", "bv"):
        assert rel_err(g1[name], g0[name]) <= 1e-5, (name, rel_err(g1[name], g0[name]))


def test_transformer_with_and_without_the_fused_pairs():
    """TwoWayTransformer.flat, one text token per bag: MIL_FUSE_LNBR=1 (default) against =0 - same outputs, same gradients."""
    from types import SimpleNamespace
    from mil_amd.model.sam.transformer import TwoWayTransformer
    lengths = [500, 1030, 260]
    B, n = len(lengths), sum(lengths)
    torch.manual_seed(0)
    tr = TwoWayTransformer(SimpleNamespace(alignment_base="CI"), depth=2, embedding_dim=512, num_heads=8, mlp_dim=2048).to(DEV)
    image = torch.randn(n, 512, device=DEV)
    point = torch.randn(B, 512, device=DEV)
    pe = ops.sinusoid_pe(max(lengths), 512, torch.device(DEV))
    dq, dk = torch.randn(B, 512, device=DEV), torch.randn(n, 512, device=DEV)
    res = {}
    for mode in ("1", "0"):
        os.environ["MIL_FUSE_LNBR"] = mode
        try:
            for p in tr.parameters():
                p.grad = None
            im, pt = image.clone().requires_grad_(True), point.clone().requires_grad_(True)
            q, k = tr.flat(im, pt, pe, lengths, [1] * B)
            ((q * dq).sum() + (k * dk).sum()).backward()
            res[mode] = (q.detach(), k.detach(), im.grad, pt.grad, {nm: p.grad.clone() for nm, p in tr.named_parameters()
                                                                    if p.grad is not None})
        finally:
            os.environ.pop("MIL_FUSE_LNBR", None)
    a, b = res["1"], res["0"]
    assert rel_err(a[0], b[0]) <= 1e-5 and rel_err(a[1], b[1]) <= 1e-5
    assert rel_err(a[2], b[2]) <= 5e-5 and rel_err(a[3], b[3]) <= 5e-5
    assert set(a[4]) == set(b[4])
    for nm in a[4]:
        assert rel_err(a[4][nm], b[4][nm]) <= 1e-4, (nm, rel_err(a[4][nm], b[4][nm]))


def test_plain_pool_backward_one_pass_matches_the_two_kernel_form(monkeypatch):
    """ops.one_token_attention (no norm in front: the first block's site): k_lnbr_apool_bwd_one<false> (default) against
    k_apool_dots + k_apool_bwd_apply (MIL_LNBR_BWD=r16), with a second consumer's gradient on the keys."""
    lengths, C = [300, 77, 512, 129], 32
    t, pe, do, dy = _inputs(lengths, C, seed=31)
    B = len(lengths)
    s_ti = AttnSegs.make([1] * B, lengths, torch.device(DEV))

    def run():
        lv = {k: t[k].clone().requires_grad_(True) for k in ("x", "qp", "Wk", "Wv", "bv")}
        o, keys = ops.one_token_attention(None, lv["x"], pe, s_ti, None, None, lv["Wk"], lv["Wv"], lv["bv"], H, qp=lv["qp"])
        ((o * do).sum() + (keys * dy).sum()).backward()
        return {k: v.grad for k, v in lv.items()}
    g1 = run()
    monkeypatch.setenv("MIL_LNBR_BWD", "r16")
    g0 = run()
    for name in g0:
        assert rel_err(g1[name], g0[name]) <= 1e-5, (name, rel_err(g1[name], g0[name]))
