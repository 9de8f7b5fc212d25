"""Seeded synthetic weights and inputs for the hot path.

There is no network (no pretrained CLIP, no hospital data), so benchmarks, tests and the
golden-vector generator all use the *architecture* of the reference with a seeded random
init.  Everything is drawn from an explicit ``torch.Generator`` on the
CPU so that the build container, the CPU test run and the GPU box rebuild bit-identical
tensors from the seed alone.  Parameter names are the reference's ``state_dict`` keys.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch

Tensor = torch.Tensor
Params = Dict[str, Tensor]

SOT_ID, EOT_ID = 49406, 49407


def _gen(seed: int) -> torch.Generator:
    return torch.Generator().manual_seed(int(seed))


def _uniform(g: torch.Generator, shape, bound: float) -> Tensor:
    return (torch.rand(shape, generator=g) * 2 - 1) * bound


def _normal(g: torch.Generator, shape, std: float) -> Tensor:
    return torch.randn(shape, generator=g) * std


def linear_params(g: torch.Generator, out_f: int, in_f: int) -> Tuple[Tensor, Tensor]:
    """nn.Linear default init: weight and bias ~ U(-1/sqrt(in), 1/sqrt(in))."""
    bound = 1.0 / math.sqrt(in_f)
    return _uniform(g, (out_f, in_f), bound), _uniform(g, (out_f,), bound)


def _put_linear(p: Params, g: torch.Generator, name: str, out_f: int, in_f: int) -> None:
    p[name + ".weight"], p[name + ".bias"] = linear_params(g, out_f, in_f)


def _put_ln(p: Params, g: torch.Generator, name: str, dim: int, jitter: float = 0.05) -> None:
    # LayerNorm defaults are ones/zeros; a small seeded jitter makes gamma/beta gradients
    # and their use observable in parity tests.
    p[name + ".weight"] = 1.0 + _uniform(g, (dim,), jitter)
    p[name + ".bias"] = _uniform(g, (dim,), jitter)


def abmil_params(seed: int = 1234, L: int = 512, D: int = 192, prefix: str = "aggregator.") -> Params:
    """model/dim1/ABMIL.py:26-36 shapes: attention_V.0 / attention_U.0 [D, L], attention_weights [1, D]."""
    g = _gen(seed)
    p: Params = {}
    _put_linear(p, g, prefix + "attention_V.0", D, L)
    _put_linear(p, g, prefix + "attention_U.0", D, L)
    _put_linear(p, g, prefix + "attention_weights", 1, D)
    return p


def head_params(seed: int = 1235, L: int = 512, C: int = 2) -> Params:
    """model/aggregator.py:128-131: fc.1 Linear(512 -> num_classes)."""
    p: Params = {}
    _put_linear(p, _gen(seed), "fc.1", C, L)
    return p


def image_only_params(seed: int = 1234, L: int = 512, D: int = 192, C: int = 2) -> Params:
    p = abmil_params(seed, L, D)
    p.update(head_params(seed + 1, L, C))
    return p


def attention_params(p: Params, g: torch.Generator, name: str, E: int, internal: int) -> None:
    """model/sam/transformer.py:413-416."""
    _put_linear(p, g, name + ".q_proj", internal, E)
    _put_linear(p, g, name + ".k_proj", internal, E)
    _put_linear(p, g, name + ".v_proj", internal, E)
    _put_linear(p, g, name + ".out_proj", E, internal)


def twoway_params(seed: int, name: str = "TwoWayTransformer_Pth", depth: int = 2, E: int = 512,
                  mlp_dim: int = 2048, downsample: int = 2) -> Params:
    """model/sam/transformer.py:39-56,260-274 parameter set."""
    g = _gen(seed)
    p: Params = {}
    for i in range(depth):
        b = f"{name}.layers.{i}"
        attention_params(p, g, b + ".self_attn", E, E)
        _put_ln(p, g, b + ".norm1", E)
        attention_params(p, g, b + ".cross_attn_token_to_image", E, E // downsample)
        _put_ln(p, g, b + ".norm2", E)
        _put_linear(p, g, b + ".mlp.lin1", mlp_dim, E)
        _put_linear(p, g, b + ".mlp.lin2", E, mlp_dim)
        _put_ln(p, g, b + ".norm3", E)
        _put_ln(p, g, b + ".norm4", E)
        attention_params(p, g, b + ".cross_attn_image_to_token", E, E // downsample)
    attention_params(p, g, name + ".final_attn_token_to_image", E, E // downsample)
    _put_ln(p, g, name + ".norm_final_attn", E)
    return p


def clip_text_params(seed: int, width: int = 512, layers: int = 12, vocab: int = 49408, ctx: int = 77,
                     embed: int = 512, prefix: str = "clinic_extractor.model.") -> Params:
    """Text-side parameters of clip/model.py:283-293 with the distributions of
    clip/model.py:295-322 (token/positional embeddings, attention/MLP stds)."""
    g = _gen(seed)
    p: Params = {}
    p[prefix + "token_embedding.weight"] = _normal(g, (vocab, width), 0.02)
    p[prefix + "positional_embedding"] = _normal(g, (ctx, width), 0.01)
    proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
    attn_std = width ** -0.5
    fc_std = (2 * width) ** -0.5
    for i in range(layers):
        b = f"{prefix}transformer.resblocks.{i}."
        p[b + "attn.in_proj_weight"] = _normal(g, (3 * width, width), attn_std)
        p[b + "attn.in_proj_bias"] = _uniform(g, (3 * width,), 0.02)
        p[b + "attn.out_proj.weight"] = _normal(g, (width, width), proj_std)
        p[b + "attn.out_proj.bias"] = _uniform(g, (width,), 0.02)
        _put_ln(p, g, b + "ln_1", width)
        p[b + "mlp.c_fc.weight"] = _normal(g, (4 * width, width), fc_std)
        p[b + "mlp.c_fc.bias"] = _uniform(g, (4 * width,), 1.0 / math.sqrt(width))
        p[b + "mlp.c_proj.weight"] = _normal(g, (width, 4 * width), proj_std)
        p[b + "mlp.c_proj.bias"] = _uniform(g, (width,), 1.0 / math.sqrt(4 * width))
        _put_ln(p, g, b + "ln_2", width)
    _put_ln(p, g, prefix + "ln_final", width)
    p[prefix + "text_projection"] = _normal(g, (width, embed), width ** -0.5)
    return p


def fused_params(seed: int = 1234, twoway: str = "TwoWayTransformer_Pth", clip_width: int = 512,
                 clip_layers: int = 12, clip_vocab: int = 49408, E: int = 512, C: int = 2, with_ct: bool = False) -> Params:
    """Live parameters of the pathology + clinical-text branch (model/aggregator.py:47,58-66,
    79-81,120-131)."""
    p: Params = {}
    g = _gen(seed + 10)
    _put_linear(p, g, "fc_pathology.0", E, 768)
    _put_linear(p, g, "fc_CI2Pth.0", E, E)
    if with_ct:
        _put_linear(p, g, "fc_CI2CT.0", E, E)                     # model/aggregator.py:44
    p.update(twoway_params(seed + 20, twoway, E=E))
    p.update(clip_text_params(seed + 30, width=clip_width, layers=clip_layers, vocab=clip_vocab, embed=E))
    p.update(image_only_params(seed, L=E, C=C))
    return p


# --------------------------------------------------------------------------- inputs
def make_bags(seed: int, B: int, N: int, L: int) -> Tensor:
    """Patch-feature bags x ~ N(0,1) fp32 [B, N, L]."""
    return torch.randn((B, N, L), generator=_gen(seed))


def make_ct_map(seed: int, B: int, D: int = 160, hw: int = 14, E: int = 512) -> Tensor:
    """A stand-in for the CT encoder's output (model/aggregator.py:139-140: (160, 512, 512) volume -> [B, 512, 160, h, w])."""
    return torch.randn((B, E, D, hw, hw), generator=_gen(seed))


def make_labels(seed: int, B: int, C: int = 2) -> Tensor:
    """One-hot float labels of randint(0, C) (dataset.py:249; train_ddp.py:293)."""
    idx = torch.randint(0, C, (B,), generator=_gen(seed))
    return torch.nn.functional.one_hot(idx, C).float()


def make_token_ids(seed: int, B: int, P: int = 1, ctx: int = 77, vocab: int = 49408) -> Tensor:
    """Synthetic tokenised notes int64 [B, P, ctx]: SOT, 20..40 ids, EOT, zero padding.

    The framing matches clip/clip.py:185-221 (SOT first, EOT last and the row maximum so
    that argmax finds it).  For a reduced vocab the two largest ids play SOT/EOT."""
    g = _gen(seed)
    sot, eot = (SOT_ID, EOT_ID) if vocab == 49408 else (vocab - 2, vocab - 1)
    ids = torch.zeros((B, P, ctx), dtype=torch.int64)
    for b in range(B):
        for q in range(P):
            n = int(torch.randint(20, 41, (1,), generator=g))
            ids[b, q, 0] = sot
            ids[b, q, 1:1 + n] = torch.randint(1, sot, (n,), generator=g)
            ids[b, q, 1 + n] = eot
    return ids


def ragged_offsets(lengths: List[int]) -> Tensor:
    """int32 [B+1] row offsets of bags concatenated along the patch axis."""
    off = torch.zeros(len(lengths) + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(torch.tensor(lengths, dtype=torch.int64), 0).to(torch.int32)
    return off
