"""Generate golden vectors by running the REFERENCE's own leaf modules on CPU.

Runs ONLY in the build container (needs /root/reference); never on the GPU box.  It
file-loads the torch-only leaf modules: model/dim1/ABMIL.py (with an empty
``torchvision`` placeholder in sys.modules: that import is unused there),
model/sam/transformer.py and clip/model.py, fills them with the seeded synthetic
weights of ``mil_amd.synthetic`` via ``load_state_dict``, wires them exactly as
model/aggregator.py:134-209 does for the pathology(+text) branch, runs eval-mode fp32
forward + BCELoss + backward, and writes inputs/outputs as small ``.npz`` files under
tests/golden/.  Fixtures hold data only - no reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

import mil_amd  # noqa: E402
from mil_amd import synthetic as syn  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def load_reference():
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tv.models)
    ab = _load("ref_abmil", os.path.join(REF, "model/dim1/ABMIL.py"))
    sys.path.insert(0, REF)
    from model.sam import transformer as tw  # namespace package, torch only
    cm = _load("ref_clip_model", os.path.join(REF, "clip/model.py"))
    return ab, tw, cm


def sub(p, prefix):
    return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}


def npz(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


SAMPLE_STRIDE = 97


def pack_grads(arrs, grads, full):
    """Every gradient is stored as its L2 norm plus a strided sample; ``full`` keeps the
    whole tensor too (small cases only, to keep tests/golden small)."""
    for k, v in grads.items():
        arrs[k + ".norm"] = v.norm()
        arrs[k + ".sample"] = v.flatten()[::SAMPLE_STRIDE]
        if full:
            arrs[k] = v


def grads_of(module, prefix=""):
    return {prefix + k: (v.grad if v.grad is not None else torch.zeros_like(v))
            for k, v in module.named_parameters()}


# --------------------------------------------------------------------------- image-only (config 1/2)
def gen_image_only(ab, tag, seed, B, N, L, store_inputs, lengths=None):
    p = syn.image_only_params(seed, L=L)
    abmil = ab.ABMIL(None, L=L).eval()
    abmil.load_state_dict(sub(p, "aggregator."))
    fc = torch.nn.Sequential(torch.nn.Dropout(0.25), torch.nn.Linear(L, 2)).eval()
    fc.load_state_dict(sub(p, "fc."))
    lengths = lengths or [N] * B
    bags = [torch.randn((n, L), generator=torch.Generator().manual_seed(seed + 100 + i)).requires_grad_(True)
            for i, n in enumerate(lengths)]
    y = syn.make_labels(seed + 7, B)
    Ms, probs, logits, scores = [], [], [], []
    for xb in bags:
        M = abmil(xb.unsqueeze(0))                       # [1, L]  (model/aggregator.py:199)
        z = fc(M)                                        # logits
        Ms.append(M); logits.append(z); probs.append(torch.sigmoid(z))   # :200
        with torch.no_grad():
            xs = xb
            A = abmil.attention_weights(abmil.attention_V(xs) * abmil.attention_U(xs))
            scores.append(A.squeeze(-1))
    prob = torch.cat(probs, 0)
    loss = torch.nn.BCELoss()(prob, y)                   # train_ddp.py:99,324
    loss.backward()
    g = grads_of(abmil, "g.aggregator.")
    g.update(grads_of(fc, "g.fc."))
    arrs = dict(seed=seed, lengths=np.array(lengths), labels=y, M=torch.cat(Ms, 0), logits=torch.cat(logits, 0),
                prob=prob, loss=loss, scores=torch.cat(scores, 0))
    dx = torch.cat([b.grad for b in bags], 0)
    pack_grads(arrs, g, full=store_inputs)
    pack_grads(arrs, {"dx": dx}, full=store_inputs)
    if store_inputs:                                     # tiny case: inputs and weights travel too
        arrs["x"] = torch.cat([b.detach() for b in bags], 0)
        for k, v in p.items():
            arrs["p." + k] = v
    npz(tag, **arrs)


def gen_quirk(ab):
    """B>1 sum-pool quirk of ABMIL.py:48,57 (documented; not the build's semantics)."""
    p = syn.image_only_params(77, L=512)
    abmil = ab.ABMIL(None, L=512).eval()
    abmil.load_state_dict(sub(p, "aggregator."))
    x = syn.make_bags(78, 3, 5, 512)
    with torch.no_grad():
        out = abmil(x)
    npz("abmil_batched_quirk", seed=77, x=x, out=out)


# --------------------------------------------------------------------------- attention / two-way
def gen_attention(tw, tag, seed, Tq, Tk, internal):
    g = torch.Generator().manual_seed(seed)
    p = {}
    syn.attention_params(p, g, "attn", 512, internal)
    m = tw.Attention(512, 8, downsample_rate=512 // internal).eval()
    m.load_state_dict(sub(p, "attn."))
    q = torch.randn((1, Tq, 512), generator=g).requires_grad_(True)
    k = torch.randn((1, Tk, 512), generator=g).requires_grad_(True)
    v = torch.randn((1, Tk, 512), generator=g).requires_grad_(True)
    out = m(q=q, k=k, v=v)
    go = torch.randn(out.shape, generator=g)
    (out * go).sum().backward()
    # inputs are re-drawn from the seed by the tests (same generator, same draw order)
    arrs = dict(seed=seed, shape=np.array([Tq, Tk, internal]), out=out[0], dq=q.grad[0], dk=k.grad[0], dv=v.grad[0])
    pack_grads(arrs, grads_of(m, "g.attn."), full=False)
    npz(tag, **arrs)


def gen_twoway(tw, tag, seed, T, N, store_inputs=True):
    name = "TwoWayTransformer_Pth"
    p = syn.twoway_params(seed, name)
    args = SimpleNamespace(alignment_base="CI", model_CT="resnetMC3_18")
    m = tw.TwoWayTransformer(args=args, depth=2, embedding_dim=512, num_heads=8, mlp_dim=2048).eval()
    m.load_state_dict(sub(p, name + "."))
    g = torch.Generator().manual_seed(seed + 1)
    img = torch.randn((1, N, 512), generator=g).requires_grad_(True)
    pt = torch.randn((1, T, 512), generator=g).requires_grad_(True)
    import oracle.mil_oracle as orc
    pe = orc.sinusoidal_pe(N, 512).unsqueeze(0)
    q, k = m(img, pe, pt)
    gq = torch.randn(q.shape, generator=g)
    gk = torch.randn(k.shape, generator=g)
    ((q * gq).sum() + (k * gk).sum()).backward()
    gr = grads_of(m, "g." + name + ".")
    # inputs are re-drawn from the seed by the tests (same generator, same draw order)
    arrs = dict(seed=seed, shape=np.array([T, N]), queries=q[0], keys=k[0], dimage=img.grad[0], dpoint=pt.grad[0])
    pack_grads(arrs, gr, full=False)
    npz(tag, **arrs)


def gen_twoway_ctmap(tw, tag, seed, T, D, hw):
    """The 5-D branch of sam/transformer.py:78-98: a CT feature map [1, 512, D, h, w] (the output of the CT encoder,
    model/aggregator.py:139-140) becomes D tokens by a mean over (h, w) and a permute; image_pe = pe[:, :D] (:160)."""
    name = "TwoWayTransformer_CT"
    p = syn.twoway_params(seed, name)
    args = SimpleNamespace(alignment_base="CI", model_CT="resnetMC3_18")
    m = tw.TwoWayTransformer(args=args, depth=2, embedding_dim=512, num_heads=8, mlp_dim=2048).eval()
    m.load_state_dict(sub(p, name + "."))
    ct = syn.make_ct_map(seed + 1, 1, D, hw)
    g = torch.Generator().manual_seed(seed + 2)
    pt = torch.randn((1, T, 512), generator=g).requires_grad_(True)
    import oracle.mil_oracle as orc
    pe = orc.sinusoidal_pe(D, 512).unsqueeze(0)
    q, k = m(ct, pe, pt)
    gq = torch.randn(q.shape, generator=g)
    gk = torch.randn(k.shape, generator=g)
    ((q * gq).sum() + (k * gk).sum()).backward()
    arrs = dict(seed=seed, shape=np.array([T, D, hw]), queries=q[0], keys=k[0], dpoint=pt.grad[0])
    pack_grads(arrs, grads_of(m, "g." + name + "."), full=False)
    npz(tag, **arrs)


def gen_twoway_ctbase(tw, tag, seed, N, D, hw):
    """`--alignment_base CT` (sam/transformer.py:78-86): the CT feature map [1, 512, D, h, w] arrives as POINT embedding and
    becomes the D query tokens (mean over (h, w), permute); image_embedding [1, N, 512] are the keys."""
    name = "TwoWayTransformer_Pth"
    p = syn.twoway_params(seed, name)
    args = SimpleNamespace(alignment_base="CT", model_CT="resnetMC3_18")
    m = tw.TwoWayTransformer(args=args, depth=2, embedding_dim=512, num_heads=8, mlp_dim=2048).eval()
    m.load_state_dict(sub(p, name + "."))
    ct = syn.make_ct_map(seed + 1, 1, D, hw)
    g = torch.Generator().manual_seed(seed + 2)
    img = torch.randn((1, N, 512), generator=g).requires_grad_(True)
    import oracle.mil_oracle as orc
    pe = orc.sinusoidal_pe(N, 512).unsqueeze(0)
    q, k = m(img, pe, ct)
    gq = torch.randn(q.shape, generator=g)
    gk = torch.randn(k.shape, generator=g)
    ((q * gq).sum() + (k * gk).sum()).backward()
    arrs = dict(seed=seed, shape=np.array([N, D, hw]), queries=q[0], keys=k[0], dimage=img.grad[0])
    pack_grads(arrs, grads_of(m, "g." + name + "."), full=False)
    npz(tag, **arrs)


def gen_fused_ct_pth(ab, tw, cm, tag, seed, B, N, P, D, hw, clip_layers):
    """model/aggregator.py:134-209 with modality ['CT', 'pathology'] and a PRECOMPUTED CT map in place of extractor_CT:
    both modalities go through TwoWayTransformer_Both (:160,168), the multi-modal bag is the 4-segment concat of :173."""
    name = "TwoWayTransformer_Both"
    p = syn.fused_params(seed, name, clip_width=512, clip_layers=clip_layers, clip_vocab=49408, with_ct=True)
    import oracle.mil_oracle as orc
    mk = lambda i, o: torch.nn.Sequential(torch.nn.Linear(i, o), torch.nn.Tanh())          # noqa: E731
    fc_path, fc_ci2p, fc_ci2c = mk(768, 512), mk(512, 512), mk(512, 512)
    fc_path.load_state_dict(sub(p, "fc_pathology."))
    fc_ci2p.load_state_dict(sub(p, "fc_CI2Pth."))
    fc_ci2c.load_state_dict(sub(p, "fc_CI2CT."))
    args = SimpleNamespace(alignment_base="CI", model_CT="resnetMC3_18")
    twm = tw.TwoWayTransformer(args=args, depth=2, embedding_dim=512, num_heads=8, mlp_dim=2048)
    twm.load_state_dict(sub(p, name + "."))
    abmil = ab.ABMIL(None, L=512)
    abmil.load_state_dict(sub(p, "aggregator."))
    fc = torch.nn.Sequential(torch.nn.Dropout(0.25), torch.nn.Linear(512, 2))
    fc.load_state_dict(sub(p, "fc."))
    clip = build_ref_clip(cm, p, 512, clip_layers, 49408, 8, 512)
    for m in (fc_path, fc_ci2p, fc_ci2c, twm, abmil, fc):
        m.eval()
    x = syn.make_bags(seed + 3, B, N, 768)
    ids = syn.make_token_ids(seed + 4, B, P)
    y = syn.make_labels(seed + 5, B)
    ct = syn.make_ct_map(seed + 6, B, D, hw)
    probs, logits, q_ct, q_p = [], [], [], []
    for b in range(B):
        with torch.no_grad():
            t = clip.encode_text(ids[b]).unsqueeze(0)
        a, c_ = twm(ct[b:b + 1], orc.sinusoidal_pe(D, 512).unsqueeze(0), fc_ci2c(t))               # :160
        e, f = twm(fc_path(x[b:b + 1]), orc.sinusoidal_pe(N, 512).unsqueeze(0), fc_ci2p(t))        # :168
        x0 = torch.cat([a, c_, e, f], dim=1)                                                      # :173
        z = fc(abmil(x0))
        logits.append(z); probs.append(torch.sigmoid(z)); q_ct.append(a[0]); q_p.append(e[0])
    prob = torch.cat(probs, 0)
    loss = torch.nn.BCELoss()(prob, y)
    loss.backward()
    arrs = dict(seed=seed, cfg=np.array([B, N, P, D, hw, clip_layers]), logits=torch.cat(logits, 0), prob=prob, loss=loss,
                x_CT2CI=torch.stack(q_ct, 0).detach(), x_Pth2CI=torch.stack(q_p, 0).detach())
    for mod, pre in ((fc_path, "fc_pathology."), (fc_ci2p, "fc_CI2Pth."), (fc_ci2c, "fc_CI2CT."), (twm, name + "."),
                     (abmil, "aggregator."), (fc, "fc.")):
        pack_grads(arrs, grads_of(mod, "g." + pre), full=False)
    npz(tag, **arrs)


def gen_twoway_block(tw, tag, seed, skip, T, N):
    p = {}
    g = torch.Generator().manual_seed(seed)
    full = syn.twoway_params(seed, "tw", depth=1)
    p = {k.replace("tw.layers.0.", "blk."): v for k, v in full.items() if k.startswith("tw.layers.0.")}
    m = tw.TwoWayAttentionBlock(512, 8, 2048, torch.nn.ReLU, 2, skip_first_layer_pe=skip).eval()
    m.load_state_dict(sub(p, "blk."))
    qs = torch.randn((1, T, 512), generator=g)
    ks = torch.randn((1, N, 512), generator=g)
    qpe = torch.randn((1, T, 512), generator=g)
    kpe = torch.randn((1, N, 512), generator=g)
    with torch.no_grad():
        q, k = m(qs, ks, qpe, kpe)
    npz(tag, seed=seed, skip=int(skip), queries_in=qs[0], keys_in=ks[0], query_pe=qpe[0], key_pe=kpe[0],
        queries=q[0], keys=k[0])


# --------------------------------------------------------------------------- CLIP text tower
def build_ref_clip(cm, p, width, layers, vocab, heads, embed):
    # vision side is built by the constructor but never run (no encode_image call on the path)
    c = cm.CLIP(embed, 32, 1, 64, 32, 77, vocab, width, heads, layers).eval()
    sd = sub(p, "clinic_extractor.model.")
    missing, unexpected = c.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("visual.") or k == "logit_scale" for k in missing), missing
    return c.float()


def gen_clip(cm, tag, seed, width, layers, vocab, heads, embed, P, store_params):
    p = syn.clip_text_params(seed, width=width, layers=layers, vocab=vocab, embed=embed)
    c = build_ref_clip(cm, p, width, layers, vocab, heads, embed)
    ids = syn.make_token_ids(seed + 1, 1, P, vocab=vocab)[0]
    with torch.no_grad():
        out = c.encode_text(ids)
    arrs = dict(seed=seed, ids=ids, out=out, cfg=np.array([width, layers, vocab, heads, embed, P]))
    if store_params:
        arrs.update({"p." + k: v for k, v in p.items()})
    npz(tag, **arrs)


def gen_coop(cm, tag, seed, width, layers, vocab, heads, embed, P, n_ctx):
    """Learnable-context branch, model/dim1/CLIP.py:32-60, run on the reference CLIP class (its tower, ln_final and
    text_projection) with gradients to ctx."""
    p = syn.clip_text_params(seed, width=width, layers=layers, vocab=vocab, embed=embed)
    c = build_ref_clip(cm, p, width, layers, vocab, heads, embed)
    for q in c.parameters():
        q.requires_grad_(False)
    x = syn.make_token_ids(seed + 1, 1, P, vocab=vocab)                  # [1, P, 77]
    g = torch.Generator().manual_seed(seed + 2)
    ctx = (torch.randn((P, n_ctx, width), generator=g) * 0.02).requires_grad_(True)
    with torch.no_grad():
        embedding = c.token_embedding(x[0, :, :]).type(c.dtype)         # CLIP.py:33
    prefix = embedding[:, :1, :]
    suffix = embedding[:, 1 + n_ctx:, :]
    prompts = torch.cat([prefix, ctx, suffix], dim=1)                    # :45-52
    x_ = prompts + c.positional_embedding.type(c.dtype)                  # :54
    x_ = x_.permute(1, 0, 2)
    x_ = c.transformer(x_)
    x_ = x_.permute(1, 0, 2)
    x_ = c.ln_final(x_).type(c.dtype)
    x_ = x_[torch.arange(x_.shape[0]), x[0].argmax(dim=-1)] @ c.text_projection     # :60
    go = torch.randn(x_.shape, generator=g)
    (x_ * go).sum().backward()
    npz(tag, seed=seed, cfg=np.array([width, layers, vocab, heads, embed, P, n_ctx]), out=x_, dctx=ctx.grad)


# --------------------------------------------------------------------------- fused recipe (config 3)
def gen_fused(ab, tw, cm, tag, seed, B, N, P, clip_layers, clip_width, clip_vocab, clip_heads):
    """model/aggregator.py:134-209, pathology + CI(text) branch, one bag per forward."""
    name = "TwoWayTransformer_Pth"
    p = syn.fused_params(seed, name, clip_width=clip_width, clip_layers=clip_layers, clip_vocab=clip_vocab)
    import oracle.mil_oracle as orc
    fc_path = torch.nn.Sequential(torch.nn.Linear(768, 512), torch.nn.Tanh())
    fc_path.load_state_dict(sub(p, "fc_pathology."))
    fc_ci = torch.nn.Sequential(torch.nn.Linear(512, 512), torch.nn.Tanh())
    fc_ci.load_state_dict(sub(p, "fc_CI2Pth."))
    args = SimpleNamespace(alignment_base="CI", model_CT="resnetMC3_18")
    twm = tw.TwoWayTransformer(args=args, depth=2, embedding_dim=512, num_heads=8, mlp_dim=2048)
    twm.load_state_dict(sub(p, name + "."))
    abmil = ab.ABMIL(None, L=512)
    abmil.load_state_dict(sub(p, "aggregator."))
    fc = torch.nn.Sequential(torch.nn.Dropout(0.25), torch.nn.Linear(512, 2))
    fc.load_state_dict(sub(p, "fc."))
    clip = build_ref_clip(cm, p, clip_width, clip_layers, clip_vocab, clip_heads, 512)
    for m in (fc_path, fc_ci, twm, abmil, fc):
        m.eval()
    x = syn.make_bags(seed + 3, B, N, 768)
    ids = syn.make_token_ids(seed + 4, B, P, vocab=clip_vocab)
    y = syn.make_labels(seed + 5, B)
    probs, logits, texts, q_out = [], [], [], []
    for b in range(B):
        xi = fc_path(x[b:b + 1])                                       # aggregator.py:149
        with torch.no_grad():                                          # dim1/CLIP.py:72-75
            t = clip.encode_text(ids[b]).unsqueeze(0)                  # [1, P, 512]
        pe = orc.sinusoidal_pe(N, 512).unsqueeze(0)                    # aggregator.py:190 (pe[:, :N])
        q, k = twm(xi, pe, fc_ci(t))
        x0 = torch.cat([q, k], dim=1)                                  # :192
        M = abmil(x0)                                                  # :199
        z = fc(M)
        logits.append(z); probs.append(torch.sigmoid(z)); texts.append(t[0]); q_out.append(q[0])   # :200
    prob = torch.cat(probs, 0)
    loss = torch.nn.BCELoss()(prob, y)
    loss.backward()
    arrs = dict(seed=seed, cfg=np.array([B, N, P, clip_layers, clip_width, clip_vocab, clip_heads]),
                logits=torch.cat(logits, 0), prob=prob, loss=loss, text=torch.stack(texts, 0),
                x_Pth2CI=torch.stack(q_out, 0).detach())
    for mod, pre in ((fc_path, "fc_pathology."), (fc_ci, "fc_CI2Pth."), (twm, name + "."),
                     (abmil, "aggregator."), (fc, "fc.")):
        pack_grads(arrs, grads_of(mod, "g." + pre), full=False)
    npz(tag, **arrs)


def gen_cossim(tag):
    """The 'textCosSim' term of the reference's loop (train_ddp.py:102,266,325-329): torch.nn.CosineEmbeddingLoss() - the
    reference calls torch's own op - on the two text-aligned tokens of the fused_ct_pth fixture (x_CT2CI, x_Pth2CI as the
    reference's TwoWayTransformer produced them), target [1]; loss and both input gradients.  A second case with random rows
    (one of them almost zero) pins the epsilon handling."""
    z = np.load(os.path.join(OUT, "fused_ct_pth.npz"))
    arrs = {}
    gen = torch.Generator().manual_seed(97)
    r1, r2 = torch.randn((5, 512), generator=gen), torch.randn((5, 512), generator=gen)
    r1[3] *= 1e-7
    for name, (a, b) in {"tok": (torch.from_numpy(z["x_CT2CI"]).squeeze(1), torch.from_numpy(z["x_Pth2CI"]).squeeze(1)),
                         "rnd": (r1, r2)}.items():
        a = a.clone().requires_grad_(True)
        b = b.clone().requires_grad_(True)
        loss = torch.nn.CosineEmbeddingLoss()(a, b, torch.tensor([1.0]))
        loss.backward()
        arrs.update({name + ".x1": a.detach(), name + ".x2": b.detach(), name + ".loss": loss.detach(),
                     name + ".dx1": a.grad, name + ".dx2": b.grad})
    npz(tag, **arrs)


def main():
    torch.set_num_threads(8)
    ab, tw, cm = load_reference()
    gen_image_only(ab, "image_only_n7", seed=11, B=1, N=7, L=512, store_inputs=True)
    gen_image_only(ab, "image_only_8x128", seed=1234, B=8, N=128, L=512, store_inputs=False)     # config 1
    gen_image_only(ab, "image_only_ragged", seed=21, B=5, N=0, L=512, store_inputs=False,
                   lengths=[1, 63, 64, 130, 257])
    gen_image_only(ab, "image_only_4x1024", seed=31, B=4, N=1024, L=512, store_inputs=False)     # config 2 shape
    gen_image_only(ab, "image_only_2x4096_L1024", seed=41, B=2, N=4096, L=1024, store_inputs=False)  # config 5 shape
    gen_quirk(ab)
    gen_attention(tw, "attn_self_T10", 51, 10, 10, 512)
    gen_attention(tw, "attn_t2i_T1_N64", 52, 1, 64, 256)
    gen_attention(tw, "attn_t2i_T10_N128", 53, 10, 128, 256)
    gen_attention(tw, "attn_i2t_N64_T10", 54, 64, 10, 256)
    gen_attention(tw, "attn_i2t_N128_T1", 55, 128, 1, 256)
    gen_twoway_block(tw, "twoway_block_skip", 61, True, 10, 64)
    gen_twoway_block(tw, "twoway_block_noskip", 62, False, 10, 64)
    gen_twoway(tw, "twoway_T1_N64", 71, 1, 64)
    gen_twoway(tw, "twoway_T10_N64", 72, 10, 64)
    gen_twoway(tw, "twoway_T1_N200", 73, 1, 200)
    gen_twoway_ctmap(tw, "twoway_ctmap_T1", 75, 1, 160, 3)
    gen_twoway_ctmap(tw, "twoway_ctmap_T10", 76, 10, 160, 2)
    gen_twoway_ctbase(tw, "twoway_ctbase_N64", 77, 64, 20, 3)
    gen_twoway_ctbase(tw, "twoway_ctbase_D160", 78, 200, 160, 2)
    gen_clip(cm, "clip_text_small", 81, width=64, layers=2, vocab=1000, heads=2, embed=64, P=3, store_params=False)
    gen_clip(cm, "clip_text_vitb32", 82, width=512, layers=12, vocab=49408, heads=8, embed=512, P=2,
             store_params=False)
    gen_coop(cm, "coop_small", 85, width=64, layers=2, vocab=1000, heads=2, embed=64, P=3, n_ctx=8)
    gen_coop(cm, "coop_w512", 86, width=512, layers=2, vocab=49408, heads=8, embed=512, P=10, n_ctx=8)
    gen_fused(ab, tw, cm, "fused_small_clip", 91, B=2, N=96, P=1, clip_layers=2, clip_width=512,
              clip_vocab=49408, clip_heads=8)
    gen_fused(ab, tw, cm, "fused_P10", 92, B=2, N=64, P=10, clip_layers=2, clip_width=512,
              clip_vocab=49408, clip_heads=8)
    gen_fused(ab, tw, cm, "fused_vitb32", 93, B=2, N=128, P=1, clip_layers=12, clip_width=512,
              clip_vocab=49408, clip_heads=8)
    gen_fused_ct_pth(ab, tw, cm, "fused_ct_pth", 95, B=2, N=96, P=1, D=160, hw=2, clip_layers=2)
    gen_cossim("cossim_ct_pth")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cossim":      # round 3: the textCosSim term on the fused_ct_pth tokens
        gen_cossim("cossim_ct_pth")
    elif len(sys.argv) > 1 and sys.argv[1] == "ct":          # only the CT-map fixtures (round 2)
        torch.set_num_threads(8)
        _ab, _tw, _cm = load_reference()
        gen_twoway_ctmap(_tw, "twoway_ctmap_T1", 75, 1, 160, 3)
        gen_twoway_ctmap(_tw, "twoway_ctmap_T10", 76, 10, 160, 2)
        gen_fused_ct_pth(_ab, _tw, _cm, "fused_ct_pth", 95, B=2, N=96, P=1, D=160, hw=2, clip_layers=2)
    elif len(sys.argv) > 1 and sys.argv[1] == "ctbase":      # round 3: --alignment_base CT
        torch.set_num_threads(8)
        _ab, _tw, _cm = load_reference()
        gen_twoway_ctbase(_tw, "twoway_ctbase_N64", 77, 64, 20, 3)
        gen_twoway_ctbase(_tw, "twoway_ctbase_D160", 78, 200, 160, 2)
    elif len(sys.argv) > 1 and sys.argv[1] == "coop":        # regenerate only the learnable-context fixtures
        torch.set_num_threads(8)
        _ab, _tw, _cm = load_reference()
        gen_coop(_cm, "coop_small", 85, width=64, layers=2, vocab=1000, heads=2, embed=64, P=3, n_ctx=8)
        gen_coop(_cm, "coop_w512", 86, width=512, layers=2, vocab=49408, heads=8, embed=512, P=10, n_ctx=8)
    else:
        main()
