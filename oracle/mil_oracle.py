"""CPU oracle for the MIL attention-pool + CLIP-text-fusion hot path.

TEST INFRASTRUCTURE ONLY.  This file restates, on torch-CPU fp32, the arithmetic of the
reference's hot path.  It is imported only by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.  The product path
(``llm-guided-multimodal-mil_amd/``) never imports it and has no CPU fallback.

Parity pinning: every function below is checked against golden vectors produced in the
build container by importing the reference's own leaf modules
(``oracle/gen_golden.py`` -> ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).
The reference ships no tests or fixtures of its own, so those vectors
are the only pin there is.  TransMIL / nystrom_attention stay "parity unpinned" and
are not restated here.

All functions are purely functional: parameters come in as a flat ``dict`` keyed by
the reference's ``state_dict`` names, one bag per call (the
reference's squeeze(0)/softmax(dim=1) is only a softmax over N when B == 1).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# --------------------------------------------------------------------------- ABMIL
def abmil_forward(x: Tensor, p: Params, prefix: str = "aggregator.",
                  keep: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """Gated-attention MIL pool.  keep=None: eval mode (dropout off).

    Follows model/dim1/ABMIL.py:47-59: V = tanh(x Wv^T + bv), U = sigmoid(x Wu^T + bu),
    s = (V*U) w^T + b, A = softmax over the N instances, M = A x.
    Train mode (ABMIL.py:26,49: x = Dropout(0.5)(x) in front of everything, so the DROPPED x is pooled too, :59): pass
    the 0/1 keep mask [N, L]; x becomes x * keep / (1 - 0.5) exactly as F.dropout scales the survivors.

    x: [N, L] one bag.  Returns (M [1, L], A [1, N], s [N]).
    """
    if keep is not None:
        x = x * keep * 2.0                           # ABMIL.py:49
    Wv, bv = p[prefix + "attention_V.0.weight"], p[prefix + "attention_V.0.bias"]
    Wu, bu = p[prefix + "attention_U.0.weight"], p[prefix + "attention_U.0.bias"]
    w, b = p[prefix + "attention_weights.weight"], p[prefix + "attention_weights.bias"]
    a_v = torch.tanh(F.linear(x, Wv, bv))            # ABMIL.py:52
    a_u = torch.sigmoid(F.linear(x, Wu, bu))         # ABMIL.py:53
    s = F.linear(a_v * a_u, w, b)                    # ABMIL.py:54  [N, 1]
    A = torch.softmax(s.transpose(-2, -1), dim=1)    # ABMIL.py:56-57  [1, N]
    M = A @ x                                        # ABMIL.py:59  [1, L]
    return M, A, s.squeeze(-1)


def abmil_forward_batched_quirk(x: Tensor, p: Params, prefix: str = "aggregator.") -> Tensor:
    """The B>1 degenerate case of ABMIL.py:48,57 (documented, NOT the build's semantics).

    With x [B>1, N, L] the squeeze(0) is a no-op and softmax(dim=1) runs over the K=1
    axis, so every weight is 1.0 and the "pool" is a plain sum over N.
    """
    Wv, bv = p[prefix + "attention_V.0.weight"], p[prefix + "attention_V.0.bias"]
    Wu, bu = p[prefix + "attention_U.0.weight"], p[prefix + "attention_U.0.bias"]
    w, b = p[prefix + "attention_weights.weight"], p[prefix + "attention_weights.bias"]
    s = F.linear(torch.tanh(F.linear(x, Wv, bv)) * torch.sigmoid(F.linear(x, Wu, bu)), w, b)
    A = torch.softmax(s.transpose(-2, -1), dim=1)    # [B, 1, N] softmax over the size-1 axis
    return A @ x                                     # [B, 1, L] == x.sum(1, keepdim=True)


# --------------------------------------------------------------------------- head + loss
def head_forward(M: Tensor, p: Params, keep: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Per-bag classifier head: z = Dropout(.25)(M) Wf^T + bf (logits), p = sigmoid(z).  keep=None: eval mode.

    model/aggregator.py:128-131,200.  M: [B, L] -> (z [B, C], prob [B, C]).  keep: 0/1 mask [B, L] of the head's dropout.
    """
    if keep is not None:
        M = M * keep * (1.0 / 0.75)                  # aggregator.py:129
    z = F.linear(M, p["fc.1.weight"], p["fc.1.bias"])
    return z, torch.sigmoid(z)


def bce_loss(prob: Tensor, y: Tensor) -> Tensor:
    """torch.nn.BCELoss(mean) on sigmoid outputs vs one-hot float labels.

    train_ddp.py:99,323-324.  log terms are clamped at -100 as torch does.
    """
    lp = torch.clamp(torch.log(prob), min=-100.0)
    l1p = torch.clamp(torch.log(1.0 - prob), min=-100.0)
    return (-(y * lp + (1.0 - y) * l1p)).mean()


def top1(prob: Tensor) -> Tensor:
    """argmax class index per bag (train_ddp.py:341-344; utils.py:159-171)."""
    return prob.argmax(dim=-1)


def accuracy(prob: Tensor, y: Tensor) -> Tensor:
    """utils.py:159-171 calculate_accuracy."""
    return (prob.argmax(dim=-1) == y.argmax(dim=-1)).float().mean()


# --------------------------------------------------------------------------- projections
def linear_tanh(x: Tensor, W: Tensor, b: Tensor) -> Tensor:
    """nn.Sequential(nn.Linear, nn.Tanh): fc_pathology (model/aggregator.py:47),
    fc_CI2CT / fc_CI2Pth / fc_CI (model/aggregator.py:44,66,68)."""
    return torch.tanh(F.linear(x, W, b))


def sinusoidal_pe(n: int, dim: int = 512) -> Tensor:
    """First n rows of the positional table of model/aggregator.py:99-106 -> [n, dim]."""
    pe = torch.zeros((n, dim))
    position = torch.arange(0, n).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, dim, 2, dtype=torch.float) * -(math.log(10000.0) / dim))
    pe[:, 0::2] = torch.sin(position.float() * div_term)
    pe[:, 1::2] = torch.cos(position.float() * div_term)
    return pe


# --------------------------------------------------------------------------- SAM two-way transformer
def layer_norm(x: Tensor, p: Params, name: str, eps: float = 1e-5) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), p[name + ".weight"], p[name + ".bias"], eps)


def attention(q: Tensor, k: Tensor, v: Tensor, p: Params, name: str, num_heads: int = 8) -> Tensor:
    """model/sam/transformer.py:428-450.  q [Tq, E], k/v [Tk, E] (one bag, no batch axis)."""
    q = F.linear(q, p[name + ".q_proj.weight"], p[name + ".q_proj.bias"])
    k = F.linear(k, p[name + ".k_proj.weight"], p[name + ".k_proj.bias"])
    v = F.linear(v, p[name + ".v_proj.weight"], p[name + ".v_proj.bias"])
    tq, internal = q.shape
    c = internal // num_heads
    qh = q.reshape(tq, num_heads, c).transpose(0, 1)             # [H, Tq, c]
    kh = k.reshape(k.shape[0], num_heads, c).transpose(0, 1)     # [H, Tk, c]
    vh = v.reshape(v.shape[0], num_heads, c).transpose(0, 1)
    attn = qh @ kh.transpose(1, 2)                               # :441
    attn = attn / math.sqrt(c)                                   # :442
    attn = torch.softmax(attn, dim=-1)                           # :443
    out = (attn @ vh).transpose(0, 1).reshape(tq, internal)      # :446-447
    return F.linear(out, p[name + ".out_proj.weight"], p[name + ".out_proj.bias"])


def mlp_block(x: Tensor, p: Params, name: str) -> Tensor:
    """model/sam/common.py:25-26 with the ReLU chosen at model/sam/transformer.py:18."""
    h = torch.relu(F.linear(x, p[name + ".lin1.weight"], p[name + ".lin1.bias"]))
    return F.linear(h, p[name + ".lin2.weight"], p[name + ".lin2.bias"])


def twoway_block(queries: Tensor, keys: Tensor, query_pe: Tensor, key_pe: Tensor, p: Params,
                 name: str, skip_first_layer_pe: bool, num_heads: int = 8) -> Tuple[Tensor, Tensor]:
    """model/sam/transformer.py:278-309."""
    if skip_first_layer_pe:                                                        # :282-283
        queries = attention(queries, queries, queries, p, name + ".self_attn", num_heads)
    else:                                                                          # :285-287
        q = queries + query_pe
        queries = queries + attention(q, q, queries, p, name + ".self_attn", num_heads)
    queries = layer_norm(queries, p, name + ".norm1")                              # :288
    q = queries + query_pe                                                         # :291-295
    k = keys + key_pe
    queries = queries + attention(q, k, keys, p, name + ".cross_attn_token_to_image", num_heads)
    queries = layer_norm(queries, p, name + ".norm2")
    queries = queries + mlp_block(queries, p, name + ".mlp")                       # :298-300
    queries = layer_norm(queries, p, name + ".norm3")
    q = queries + query_pe                                                         # :303-307
    k = keys + key_pe
    keys = keys + attention(k, q, queries, p, name + ".cross_attn_image_to_token", num_heads)
    keys = layer_norm(keys, p, name + ".norm4")
    return queries, keys


def twoway_transformer(image: Tensor, image_pe: Tensor, point: Tensor, p: Params, name: str,
                       depth: int = 2, num_heads: int = 8) -> Tuple[Tensor, Tensor]:
    """model/sam/transformer.py:100-120 for already-tokenised inputs.

    image [N, E] (keys), image_pe [N, E], point [T, E] (text tokens = queries and query_pe).
    Returns (queries [T, E], keys [N, E]).
    """
    queries, keys = point, image
    for i in range(depth):
        queries, keys = twoway_block(queries, keys, point, image_pe, p, f"{name}.layers.{i}",
                                     skip_first_layer_pe=(i == 0), num_heads=num_heads)
    q = queries + point                                                            # :114-118
    k = keys + image_pe
    queries = queries + attention(q, k, keys, p, name + ".final_attn_token_to_image", num_heads)
    queries = layer_norm(queries, p, name + ".norm_final_attn")
    return queries, keys


# --------------------------------------------------------------------------- CLIP text tower
def clip_causal_mask(ctx: int) -> Tensor:
    """clip/model.py:324-330: additive mask, -inf strictly above the diagonal."""
    return torch.full((ctx, ctx), float("-inf")).triu_(1)


def _clip_tower(x: Tensor, eot: Tensor, p: Params, heads: int, prefix: str) -> Tensor:
    """Blocks of clip/model.py:167-199 (pre-LN, nn.MultiheadAttention with the additive causal mask, QuickGELU MLP),
    ln_final, the row at the EOT position, text_projection.  x [P, ctx, W] already holds token + positional embeddings."""
    P_, ctx, width = x.shape
    c = width // heads
    mask = clip_causal_mask(ctx)
    n_layers = 1 + max(int(k[len(prefix + "transformer.resblocks."):].split(".")[0])
                       for k in p if k.startswith(prefix + "transformer.resblocks."))
    for i in range(n_layers):
        b = f"{prefix}transformer.resblocks.{i}."
        h = F.layer_norm(x, (width,), p[b + "ln_1.weight"], p[b + "ln_1.bias"], 1e-5)
        qkv = F.linear(h, p[b + "attn.in_proj_weight"], p[b + "attn.in_proj_bias"])
        q, k, v = qkv.split(width, dim=-1)
        q = q.reshape(P_, ctx, heads, c).transpose(1, 2)                   # [P, H, ctx, c]
        k = k.reshape(P_, ctx, heads, c).transpose(1, 2)
        v = v.reshape(P_, ctx, heads, c).transpose(1, 2)
        att = torch.softmax((q / math.sqrt(c)) @ k.transpose(-2, -1) + mask, dim=-1)
        o = (att @ v).transpose(1, 2).reshape(P_, ctx, width)
        x = x + F.linear(o, p[b + "attn.out_proj.weight"], p[b + "attn.out_proj.bias"])
        h = F.layer_norm(x, (width,), p[b + "ln_2.weight"], p[b + "ln_2.bias"], 1e-5)
        h = F.linear(h, p[b + "mlp.c_fc.weight"], p[b + "mlp.c_fc.bias"])
        h = h * torch.sigmoid(1.702 * h)                                   # clip/model.py:162-164
        x = x + F.linear(h, p[b + "mlp.c_proj.weight"], p[b + "mlp.c_proj.bias"])
    x = F.layer_norm(x, (width,), p[prefix + "ln_final.weight"], p[prefix + "ln_final.bias"], 1e-5)
    return x[torch.arange(P_), eot] @ p[prefix + "text_projection"]


def clip_encode_text(ids: Tensor, p: Params, heads: int, prefix: str = "clinic_extractor.model.") -> Tensor:
    """clip/model.py:339-352 encode_text in fp32: ids int64 [P, ctx] -> [P, embed].  The feature is the ln_final row
    at argmax(ids) (the EOT token has the largest id) times text_projection."""
    x = p[prefix + "token_embedding.weight"][ids] + p[prefix + "positional_embedding"]
    return _clip_tower(x, ids.argmax(dim=-1), p, heads, prefix)


def clip_learnable_prompts(ids: Tensor, ctx_vectors: Tensor, p: Params, heads: int,
                           prefix: str = "clinic_extractor.model.") -> Tensor:
    """Learnable-context (CoOp) branch, model/dim1/CLIP.py:32-60, for the prompts of ONE bag: ids [P, ctx],
    ctx_vectors [P, n_ctx, W] replace token positions 1..n_ctx; the tower stays frozen, gradients reach ctx_vectors."""
    n_ctx = ctx_vectors.shape[1]
    emb = p[prefix + "token_embedding.weight"][ids].detach()               # :33
    prompts = torch.cat([emb[:, :1, :], ctx_vectors, emb[:, 1 + n_ctx:, :]], dim=1)      # :45-52
    x = prompts + p[prefix + "positional_embedding"]                       # :54
    return _clip_tower(x, ids.argmax(dim=-1), p, heads, prefix)            # :55-60


# --------------------------------------------------------------------------- full recipes
def image_only_forward(x: Tensor, p: Params, keep_x: Optional[Tensor] = None,
                       keep_m: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """BASELINE config 2 for one bag x [N, L=512]: ABMIL -> fc -> sigmoid
    (model/aggregator.py:199-200 with the bag fed directly).  keep_x [N, L] / keep_m [1, L]: the
    dropout masks of a model.train() pass (None = eval)."""
    M, A, s = abmil_forward(x, p, keep=keep_x)
    z, prob = head_forward(M, p, keep=keep_m)
    return {"M": M, "A": A, "scores": s, "logits": z, "prob": prob}


def fused_forward(x768: Tensor, ids: Tensor, p: Params, clip_heads: int = 8,
                  twoway: str = "TwoWayTransformer_Pth") -> Dict[str, Tensor]:
    """BASELINE config 3 for one bag, pathology + clinical-text branch of
    model/aggregator.py:134-209 (:149, :151, :190-192, :199-200).

    x768 [N, 768] patch features, ids int64 [P, ctx] tokenised note(s).
    """
    xi = linear_tanh(x768, p["fc_pathology.0.weight"], p["fc_pathology.0.bias"])         # :149
    with torch.no_grad():                                                                 # dim1/CLIP.py:72
        t = clip_encode_text(ids, p, clip_heads)                                          # :151
    point = linear_tanh(t, p["fc_CI2Pth.0.weight"], p["fc_CI2Pth.0.bias"])
    pe = sinusoidal_pe(xi.shape[0], xi.shape[1])
    q, k = twoway_transformer(xi, pe, point, p, twoway)                                   # :190
    x0 = torch.cat([q, k], dim=0)                                                         # :192
    M, A, s = abmil_forward(x0, p)                                                        # :199
    z, prob = head_forward(M, p)                                                          # :200
    return {"text": t, "x_Pth2CI": q, "x_CI2Pth": k, "M": M, "A": A, "scores": s,
            "logits": z, "prob": prob}


def ct_map_tokens(ct: Tensor, model_CT: str = "resnetMC3_18") -> Tensor:
    """sam/transformer.py:86-98: a CT feature map [B, 512, D, h, w] becomes tokens [B, D, 512] by a mean over (h, w) and a
    permute (resnetMC3_18), or [B, D*h*w, 512] by flatten(2).permute (medicalNet)."""
    if model_CT == "medicalNet":
        return ct.flatten(2).permute(0, 2, 1)
    return ct.mean(dim=(3, 4)).permute(0, 2, 1)


def fused_forward_ct_pth(ct: Tensor, x768: Tensor, ids: Tensor, p: Params, clip_heads: int = 8) -> Dict[str, Tensor]:
    """model/aggregator.py:134-209 for one patient with modality ['CT', 'pathology'] and a precomputed CT map
    ct [512, D, h, w] in place of extractor_CT's output: both modalities through TwoWayTransformer_Both (:160,168), the
    multi-modal bag is cat([x_CT2CI, x_CI2CT, x_Pth2CI, x_CI2Pth]) (:173)."""
    tw = "TwoWayTransformer_Both"
    with torch.no_grad():
        t = clip_encode_text(ids, p, clip_heads)                                             # :151
    ctk = ct_map_tokens(ct.unsqueeze(0))[0]                                                   # [D, 512]
    a, c_ = twoway_transformer(ctk, sinusoidal_pe(ctk.shape[0], 512), linear_tanh(t, p["fc_CI2CT.0.weight"], p["fc_CI2CT.0.bias"]), p, tw)   # :160
    xi = linear_tanh(x768, p["fc_pathology.0.weight"], p["fc_pathology.0.bias"])             # :141
    e, f = twoway_transformer(xi, sinusoidal_pe(xi.shape[0], 512), linear_tanh(t, p["fc_CI2Pth.0.weight"], p["fc_CI2Pth.0.bias"]), p, tw)    # :168
    x0 = torch.cat([a, c_, e, f], dim=0)                                                      # :173
    M, A, s = abmil_forward(x0, p)
    z, prob = head_forward(M, p)
    return {"x_CT2CI": a, "x_Pth2CI": e, "M": M, "logits": z, "prob": prob}


def batch_loss_and_grads(bags: List[Tensor], labels: Tensor, p: Params, forward=image_only_forward,
                         wrt: Optional[List[str]] = None, extra: Optional[List[Tensor]] = None):
    """Mean-BCE over a list of independent B=1 bags, and its gradients (torch autograd on
    the restated ops, which is exactly what the reference's loss.backward() runs).

    Returns (loss, logits [B, C], prob [B, C], grads dict).
    """
    names = wrt if wrt is not None else [k for k in p if p[k].is_floating_point()]
    leaves = {k: p[k].detach().clone().requires_grad_(True) for k in names}
    q = dict(p)
    q.update(leaves)
    outs = []
    for i, xb in enumerate(bags):
        outs.append(forward(xb, q) if extra is None else forward(xb, extra[i], q))
    prob = torch.cat([o["prob"] for o in outs], 0)
    logits = torch.cat([o["logits"] for o in outs], 0)
    loss = bce_loss(prob, labels)
    used = [k for k in names]
    g = torch.autograd.grad(loss, [leaves[k] for k in used], allow_unused=True)
    grads = {k: (gi if gi is not None else torch.zeros_like(leaves[k])) for k, gi in zip(used, g)}
    return loss.detach(), logits.detach(), prob.detach(), grads


def cosine_embedding_loss(x1: Tensor, x2: Tensor) -> Tensor:
    """criterion_CosSim(train_CI[0].squeeze(1), train_CI[1].squeeze(1), label_for_CosSim = [1]) of the reference's loop
    (train_ddp.py:102,266,325-329): torch.nn.CosineEmbeddingLoss with target +1, mean reduction, restated from ATen's
    cosine_embedding_loss (EPSILON = 1e-12 added to both squared norms)."""
    dot = (x1 * x2).sum(1)
    m1 = (x1 * x1).sum(1) + 1e-12
    m2 = (x2 * x2).sum(1) + 1e-12
    return (1.0 - dot / torch.sqrt(m1 * m2)).mean()


# --------------------------------------------------------------------------- CLIP-as-loss
def clip_contrastive_loss(out: Tensor, feat: Tensor) -> Tensor:
    """utils.py:276-282 (CLIPloss_v1.forward after the text features are built): out [b, E], feat [b, F, E].
    logits [F, b, b] = out[None] @ feat.permute(1, 2, 0); identity targets repeated over F; CrossEntropyLoss
    with probability targets of shape [F, b, b] (class axis = dim 1)."""
    logits = torch.matmul(out.unsqueeze(0), feat.permute(1, 2, 0))
    labels = torch.eye(out.shape[0]).unsqueeze(0).repeat(logits.shape[0], 1, 1)
    return F.cross_entropy(logits, labels)


# --------------------------------------------------------------------------- data-parallel partition
def distributed_sampler_indices(n: int, world: int, rank: int, epoch: int = 0, shuffle: bool = True,
                                seed: int = 0) -> List[int]:
    """Index list torch.utils.data.DistributedSampler yields (train_ddp.py:191,201):
    seeded permutation (seed + epoch), wrap-around padding to a multiple of world,
    rank r takes indices r::world."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    total = int(math.ceil(n / world)) * world
    pad = total - len(idx)
    if pad > 0:
        idx += (idx * int(math.ceil(pad / len(idx))))[:pad]
    return idx[rank:total:world]


def adam_step(param: Tensor, grad: Tensor, m: Tensor, v: Tensor, step: int, lr: float = 1e-5,
              b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8, wd: float = 1e-7):
    """torch.optim.Adam (L2 weight decay folded into the gradient), train_ddp.py:115-118."""
    g = grad + wd * param
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mhat = m / (1 - b1 ** step)
    vhat = v / (1 - b2 ** step)
    return param - lr * mhat / (vhat.sqrt() + eps), m, v
